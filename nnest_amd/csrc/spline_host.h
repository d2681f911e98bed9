// spline_host.h -- the spline handle, shared by nnest_spline.hip (inference, image build) and nnest_spline_train.hip
#pragma once
#include <vector>
#include "nnest_internal.h"

struct nnest_spline {
    nnest::SplineShape s;
    int device, num_cu;
    std::vector<float> w;     // packed weights, state_dict order (host master copy)
    std::vector<float> perm;  // B x D x D permutation matrices P (fixed, not part of the state_dict)
    std::vector<float> img_host;
    float *img;               // device fragment image (inference: folded affine maps, both directions)
    // training state (allocated on first use, nnest_spline_train.hip)
    float *w_dev, *adam_m, *adam_v, *best_w;  // packed, device
    int *pi_dev;              // [B][D]: column of the 1 in row i of P
    int *pos_dev;             // [2][num_params]: packed conditioner parameter -> element of the forward / transposed training image
    float *wmat;              // [B][D][D] assembled W
    float *timg;              // training image
    float *partial;           // per-wave gradient / loss slices
    float *grad;              // reduced packed gradient
    float *gwsum;             // reduced dLoss/dW of the convs [B][D][D]
    float *stash;             // block inputs of the forward pass
    float *keep;              // activations and spline parameters the forward pass keeps for the backward pass
    float *gbuf, *hbuf;       // per coupling: dLoss/d(raw spline parameters) and the last hidden activations of every row (spl_w3_*)
    float *losses_dev;        // per-step losses of an epoch + validation
    void *ctl_dev;            // SplTrainCtl: the early-stopping state of a training call (kept on the device)
    void *ctl_host;           // pinned: two snapshots of it
    float *epoch_losses_dev;  // [2 * epoch_losses_cap]: train / validation loss per epoch
    int epoch_losses_cap;
    int partial_tiles;
    int adam_step;
    bool w_dev_current;       // w_dev holds the same weights as w
};

namespace nnest {
int spline_fail(int code, const char *fmt, ...);
int spline_build_image(nnest_spline *h);
int spline_mlp_params(int nin, int nout, int H);
}  // namespace nnest

#define SHIP_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) return nnest::spline_fail(NNEST_E_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)
