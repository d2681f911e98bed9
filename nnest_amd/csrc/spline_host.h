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
    void *rows;               // SplRowsState: the rows form's staging buffers, W^T and log-det constants (nnest_spline_rows.hip)
};

namespace nnest {
enum { SPL_W_SLACK_BYTES = 64 * 1024 };   // zeroed bytes behind w_dev (nnest_spline_train.hip: ensure_train_state)
struct SplTrainShape;
// ---- the rows form of the training step (nnest_spline_rows.hip) ----
struct SplRowsBatch {   // one gradient launch: a minibatch (+ Mv forward-only validation rows behind it)
    const float *x;
    const int *perm;
    int M, mtot;
    const float *noise;
    uint64_t seed;
    long noise_row0;
    int epoch;
    float jitter;
    const float *xv;
    int Mv;
    const int *stop;
};
struct SplRowsStep {    // one update launch
    int M;
    float step_size, inv_bc2s, wd, ldw;
    float *loss_out;
    float loss_scale;
    const int *stop;
    float *grad_out, *gwsum_out;   // both non-NULL: write the gradient instead of stepping
};
bool spline_rows_eligible(const SplineShape &s, int batch);
int spline_rows_prepare(nnest_spline *h, const SplTrainShape &ts, int max_rows, int valid_rows, hipStream_t st, const int *stop);
hipError_t spline_rows_grad(nnest_spline *h, const SplTrainShape &ts, const SplRowsBatch &bt, hipStream_t st);
hipError_t spline_rows_update(nnest_spline *h, const SplTrainShape &ts, const SplRowsStep &u, hipStream_t st);
float *spline_rows_rowlp(nnest_spline *h);
void spline_rows_free(nnest_spline *h);
int spline_fail(int code, const char *fmt, ...);
int spline_build_image(nnest_spline *h);
int spline_mlp_params(int nin, int nout, int H);
}  // namespace nnest

#define SHIP_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) return nnest::spline_fail(NNEST_E_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)
