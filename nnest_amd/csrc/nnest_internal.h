// nnest_internal.h -- C++ declarations shared by the .hip translation units of libnnest_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flow_tile.h"
#include "maf_tile.h"
#include "spline_tile.h"
#include "../../include/nnest_hip.h"

namespace nnest {

enum { PASS_FORWARD = 0, PASS_INVERSE = 1, PASS_LOGPROB = 2, PASS_INVERSE_LOGLIKE = 3 };

void set_last_error(const char *msg);  // nnest_abi.hip: the string behind nnest_hip_last_error()
bool shape_supported(const FlowShape &s);
hipError_t launch_repack(const float *packed, float *img, const FlowShape &s, hipStream_t st);
hipError_t launch_zero_scale_nets(float *packed, const FlowShape &s, hipStream_t st);
hipError_t launch_pass(const float *img, const FlowShape &s, int mode, const float *in, float *out, float *logdet,
                       double *logl, int *inbox, int N, const LikeSpec &like, int num_cu, hipStream_t st);
hipError_t launch_mh(const float *img, const FlowShape &s, const LikeSpec &like, float *z, float *x, double *logl,
                     double loglstar, float step_size, int steps, int C, int flags, const float *noise_dz,
                     const float *noise_u, uint64_t seed, uint64_t walker_offset, float *hist_x, double *hist_logl,
                     int *n_accept, int *n_call, float *scale_out, const float *packed, unsigned long long *sync, int num_cu,
                     hipStream_t st);
struct MhArgs;
bool quad_form_eligible(const MhArgs &a, int num_cu);        // nnest_quad.hip
hipError_t launch_mh_quad(const MhArgs &a, int num_cu, hipStream_t st);  // nnest_quad.hip
bool solo_form_eligible(const MhArgs &a, int num_cu);        // nnest_solo.hip
hipError_t launch_mh_solo(const MhArgs &a, int num_cu, hipStream_t st);  // nnest_solo.hip
int mh_form_for(const FlowShape &s, int C, int flags, int num_cu);
bool maf_shape_supported(const FlowShape &s);   // maf_kernels.h (nnest_kernels.hip)
hipError_t launch_maf_repack(const float *packed, float *imgf, float *imgb, const FlowShape &s, hipStream_t st);
hipError_t launch_loglike(const LikeSpec &like, const float *x, double *logl, int N, int D, int num_cu, hipStream_t st);
hipError_t launch_fill_noise(float *dz, float *u, int steps, int C, int D, uint64_t seed, uint64_t walker_offset,
                             hipStream_t st);
int mh_num_groups(int C);

// neural-spline flow (spline_kernels.h inside nnest_kernels.hip; the pair form of the proposal kernel in nnest_spline_mh.hip; host
// side in nnest_spline.hip)
struct SplArgs {
    const float *img;
    SplineShape sp;
};
hipError_t launch_spline_mh_pair(const MhArgs &a, const SplArgs &q, bool dbg, hipStream_t st);   // nnest_spline_mh.hip
hipError_t launch_spline_mh_team(const MhArgs &a, const SplArgs &q, bool dbg, hipStream_t st);   // nnest_spline_mh.hip
int spline_mh_form(const SplineShape &sp, int C, int flags, int num_cu);   // spline_kernels.h
bool spline_shape_supported(const SplineShape &s);
hipError_t launch_spline_pass(const float *img, const SplineShape &sp, int mode, const float *in, float *out, float *logdet,
                              double *logl, int *inbox, int N, const LikeSpec &like, int num_cu, hipStream_t st);
hipError_t launch_spline_mh(const float *img, const SplineShape &sp, const LikeSpec &like, float *z, float *x, double *logl,
                            double loglstar, float step_size, int steps, int C, int flags, const float *noise_dz,
                            const float *noise_u, uint64_t seed, uint64_t walker_offset, float *hist_x, double *hist_logl,
                            int *n_accept, int *n_call, float *scale_out, unsigned long long *sync, int num_cu, hipStream_t st);

// training (nnest_train.hip)
struct TrainArgs;
hipError_t launch_loss_grad(const float *packed, const FlowShape &s, const float *x, int M, float *grad, float *loss,
                            float *workspace, float *img_fwd, const int *fwd_pos, const int *bwd_pos, hipStream_t st);
hipError_t launch_vjp(const float *packed, const FlowShape &s, const float *x, const float *gz, float gld, int M, float *grad, float *gx,
                      float *workspace, float *img_fwd, const int *fwd_pos, const int *bwd_pos, hipStream_t st);
hipError_t launch_adam_packed(float *w, const float *grad, float *m, float *v, int n, int step, float lr, float wd, hipStream_t st);
hipError_t launch_adam_packed_dev(float *w, const float *grad, float *m, float *v, int n, int *step_dev, float lr, float wd, hipStream_t st);
size_t train_workspace_floats(const FlowShape &s, int batch);
hipError_t launch_train(float *packed, float *adam_m, float *adam_v, float *best_w, float *img, int *adam_step_dev,
                        const FlowShape &s, const float *xtrain, int n_train, const float *xvalid, int n_valid,
                        const int *perm, const float *noise, uint64_t seed, float jitter, int batch, int max_epochs,
                        int patience, float lr, float wd, int epoch_offset, int flags, float *losses, nnest_train_result_t *result,
                        float *workspace, const int *fwd_pos, const int *bwd_pos,
                        hipStream_t st);
hipError_t launch_build_pos(int *fwd_pos, int *bwd_pos, const FlowShape &s, hipStream_t st);
// masked autoregressive flow (maf_train.h inside nnest_train.hip)
size_t maf_workspace_floats(const FlowShape &s);
hipError_t launch_maf_build_gpos(int *gpos, const FlowShape &s, hipStream_t st);
hipError_t launch_maf_build_pos(int *fwd_pos, int *bwd_pos, const FlowShape &s, hipStream_t st);
hipError_t launch_maf_train_minibatch(const FlowShape &s, float *imgf, float *imgb, const int *gpos, const int *fwd_pos, const int *bwd_pos,
                                      const float *x, int M, float *w, float *m, float *v, int *step_dev, float lr, float wd,
                                      float *loss_acc, unsigned int *ticket, float *workspace, hipStream_t st);
hipError_t launch_maf_loss_grad(const FlowShape &s, const float *imgf, const float *imgb, const int *gpos, const float *x, int M,
                                float *grad, float *loss, float *workspace, hipStream_t st);
hipError_t launch_training_jitter(const double *samples, int N, int D, double *out, hipStream_t st);
// slice proposal in latent space (nnest_solo.hip; build-defined: the reference has none)
bool slice_form_eligible(const FlowShape &s);
hipError_t launch_slice_solo(const FlowShape &s, const float *packed, const LikeSpec &like, float *z, float *x, double *logl, double loglstar,
                             float width, int steps, int C, int max_out, int max_shrink, uint64_t seed, uint64_t walker_offset,
                             const float *noise_dz, float *hist_x, int *n_call, int *n_move, int *n_eval, hipStream_t st);
hipError_t launch_slice_fill_noise(float *dz, int steps, int C, int D, uint64_t seed, uint64_t walker_offset, hipStream_t st);

}  // namespace nnest
