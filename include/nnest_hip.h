/*
 * nnest_hip.h -- C ABI of libnnest_hip.so: the MI355X (gfx950) implementation of the nnest
 * flow-transform + batched-proposal + likelihood + flow-training hot path.
 *
 * The reference (adammoss/nnest v0.4.2) is pure Python and has no FFI; the seam it offers is
 * constructor injection of a Trainer-shaped object (nnest/sampler.py:50, :196-212;
 * nnest/nested.py:44, :83).  Each entry point below names the reference function it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - all pointers named *_dev are DEVICE pointers (e.g. torch tensor .data_ptr()); the caller owns
 *     every buffer; rows are row-major [N, D] float32 unless stated
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous
 *     on that stream unless documented otherwise; nothing here calls hipDeviceSynchronize
 *   - return value: 0 = ok, non-zero = error (NNEST_E_*); message via nnest_hip_last_error();
 *     nothing throws across the ABI
 *   - packed weights are float32 in torch state_dict order: per block,
 *     scale_net {W[H,D] b[H] (W[H,H] b[H])xL W[D,H] b[D]} then translate_net (same shapes)
 */
#ifndef NNEST_HIP_H
#define NNEST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNEST_HIP_ABI_VERSION 15

enum {
    NNEST_OK = 0,
    NNEST_E_ARG = 1,         /* bad argument / unsupported configuration */
    NNEST_E_HIP = 2,         /* HIP runtime error */
    NNEST_E_UNSUPPORTED = 3, /* shape outside what the kernels are instantiated for */
};

/* likelihood ids for the fused kernels (nnest/likelihoods.py) */
enum {
    NNEST_LIKE_ROSENBROCK = 0, /* Rosenbrock.loglike   likelihoods.py:51 */
    NNEST_LIKE_GAUSSMIX = 1,   /* GaussianMix.loglike  likelihoods.py:165-189 (sep 4, sigma 1, w .4 .3 .2 .1) */
    NNEST_LIKE_HIMMELBLAU = 2, /* Himmelblau.loglike   likelihoods.py:70 (D>2: sum over consecutive pairs) */
    NNEST_LIKE_GAUSSIAN = 3,   /* Gaussian.loglike     likelihoods.py:77-94  params[0] = corr (equicorrelated covariance) */
    NNEST_LIKE_EGGBOX = 4,     /* Eggbox.loglike       likelihoods.py:97-110 (x_dim = 2) */
    NNEST_LIKE_SHELL = 5,      /* GaussianShell        likelihoods.py:113-132 params = sigma, rshell, center */
    NNEST_LIKE_DOUBLE_SHELL = 6, /* DoubleGaussianShell likelihoods.py:135-150 params = sigma1, rshell1, center1,
                                    sigma2, rshell2, center2 (weights 1, 1) */
    NNEST_LIKE_COUNT = 7
};

/* which analytic likelihood the fused kernels evaluate, and on what: logl = loglike(scale * x)
 * (the reference's transform = lambda x: scale * x, examples/nested/run.py:25-42).  Host struct. */
typedef struct {
    int id;          /* NNEST_LIKE_* */
    float scale;
    float params[6]; /* per-likelihood parameters, see the enum; unused entries ignored */
} nnest_like_t;

/* flags for nnest_mh_constrained_steps */
enum {
    NNEST_MH_DYNAMIC_STEP = 1, /* sampler.py:422-431 step-size adaptation applied per group of 16 walkers (one wave): equals
                                * the reference's rule for batches of <= 16 chains, shard-invariant, no cross-workgroup
                                * traffic */
    NNEST_MH_UNCONSTRAINED = 2, /* loglstar = None (sampler.py:371-410): plain Metropolis with the likelihood and the box
                                 * prior in the ratio, min(1, exp(dlogdet + dlogl)); `loglstar` is ignored and every
                                 * proposal counts as one likelihood call */
    NNEST_MH_DYNAMIC_BATCH = 4, /* sampler.py:422-431 over ALL C walkers of the launch, as the reference applies it: every
                                 * workgroup posts its accepted count per step to `sync_dev`; the scale used from step
                                 * s + 1 + lag on reflects the batch-wide count of step s.  lag = flags bits 8..11
                                 * (NNEST_MH_LAG): 0 is the reference's rule exactly (a grid-wide wait per step), lag >= 1
                                 * takes the wait off the step's critical path (the counts are requested one step ahead).
                                 * Needs every workgroup resident: NNEST_E_UNSUPPORTED beyond ~16 walkers x 8 x CUs */
};
#define NNEST_MH_LAG(n) (((n) & 15) << 8)
/* flags bits 29 / 30: sync_dev is one half of a double buffer whose other half lies BEHIND it (bit 29) / IN FRONT of it (bit 30),
 * nnest_mh_sync_words(steps) words away, and the launch zeroes that other half for the next launch (see sync_dev below) */
#define NNEST_MH_SYNC_ZERO_NEXT (1 << 29)
#define NNEST_MH_SYNC_ZERO_PREV (1 << 30)
/* flags bits 20..27, with NNEST_MH_DYNAMIC_BATCH and lag >= 1: the first n steps of the launch apply the rule EXACTLY (lag 0, a
 * grid-wide wait on each of them) and only the steps after them run `lag` behind: votes 1..n are applied as the reference
 * applies them, the votes of the steps s > n from step s + 1 + lag on.  The rule's gain is 1 / (1 + votes) and every launch
 * starts from the caller's step_size, so the early votes are the ones that move the scale.  Kernel forms that do not implement
 * it return NNEST_E_UNSUPPORTED (nnest_mh_form_for tells). */
#define NNEST_MH_WARM(n) (((n) & 255) << 20)
/* flags bits 16..19: pin the kernel form (0 = by population).  A caller that shards ONE batch over ranks pins the form the
 * whole batch would get, so that a shard reproduces the slice of the unsharded run bit for bit. */
enum { NNEST_MH_FORM_AUTO = 0, NNEST_MH_FORM_IMAGE = 1, NNEST_MH_FORM_REG = 2, NNEST_MH_FORM_TEAM = 3, NNEST_MH_FORM_QUAD = 4,
       NNEST_MH_FORM_QUAD1 = 5, /* the quad tile with both nets on one wave (same bits as QUAD; A/B diagnostic) */
       NNEST_MH_FORM_SOLO = 6   /* one walker per wave, layers as v_fmac_f32 + DPP row rotations (nnest_solo.hip): <= 4 walkers per CU,
                                 * x_dim <= 128 (beyond 64 with the weights in LDS), fixed step or the batch-wide rule at lag >= 3 or at lag 0
                                 * (round 5: the reference's rule itself, every step an exact step) */ };
#define NNEST_MH_FORM(f) (((f) & 15) << 16)

typedef struct nnest_nvp nnest_nvp_t; /* opaque: RealNVP coupling stack + Adam state on one device */

int nnest_hip_version(void);
const char *nnest_hip_last_error(void);
/* number of compute units / device name of the current device (diagnostics for bench.py) */
int nnest_hip_device_info(int *num_cu, int *clock_khz, char *name, int name_len);

/* SingleSpeedNVP(num_inputs=D, num_hidden=H, num_blocks=B, num_layers=L) -- networks.py:328-347.
 * Allocates device storage for the packed weights, Adam moments and the MFMA-fragment image. */
int nnest_nvp_create(int D, int H, int B, int L, nnest_nvp_t **out);
/* SingleSpeedNVP(..., scale=) -- networks.py:328-347: '' (full affine coupling), 'translate'
 * (CouplingLayer(translate_only=True), networks.py:293-294, :304-305) or 'constant' (translate-only couplings, each
 * followed by a ScaleLayer, networks.py:312-325: y = x e^s, logdet += s).
 * Packed layout for every mode: the B blocks as for '' (scale_net then translate_net); with 'translate' and
 * 'constant' the scale_net slots are unused -- forced to zero on load, never given a gradient -- and with
 * 'constant' the B ScaleLayer scalars follow the blocks (nnest_nvp_num_params counts them). */
enum { NNEST_SCALE_AFFINE = 0, NNEST_SCALE_TRANSLATE = 1, NNEST_SCALE_CONSTANT = 2 };
int nnest_nvp_create_scaled(int D, int H, int B, int L, int scale_mode, nnest_nvp_t **out);
/* Masked autoregressive flow (named by BASELINE config 5).  ABSENT FROM THE REFERENCE (nnest/trainer.py:83-100
 * dispatches 'choleksy' / 'nvp' / 'spline' only): build-defined, DESIGN.md 3c -- B blocks of two MADE-masked nets with the
 * shapes of the coupling nets (so the packed vector has the SingleSpeedNVP layout and size), dimension order reversed between
 * blocks; forward (x -> z, the density / training direction) is one pass, inverse (z -> x, sampling and the Metropolis
 * proposals) runs group by group (nnest_maf_num_groups passes per block, at most hidden_dim + 1).  The handle is an
 * nnest_nvp_t: every nnest_nvp_* entry point (weights, Adam state, forward / inverse / log_probs / inverse_loglike, loss_grad,
 * adam_step) and nnest_mh_constrained_steps accept it; nnest_nvp_train and nnest_nvp_vjp return NNEST_E_UNSUPPORTED (the epoch
 * loop of this flow is driven from the host).  hidden_dim 16, x_dim 2..128. */
int nnest_maf_create(int D, int H, int B, int L, nnest_nvp_t **out);
int nnest_maf_num_groups(const nnest_nvp_t *maf);
int nnest_nvp_destroy(nnest_nvp_t *nvp);
int nnest_nvp_num_params(const nnest_nvp_t *nvp);
/* Base distribution of the flow (NormalizingFlowModel(prior=...), networks.py:47-59; Trainer(base_dist=...), trainer.py:41):
 * beta = 0 (default) is MultivariateNormal(0, I); beta > 0 is the reference's GeneralisedNormal(loc 0, scale 1, beta)
 * (nnest/distributions/generalised_normal.py:66-71; examples/nested/run.py --base_dist gen_normal --beta 8).  It enters
 * log_probs and therefore the training loss; forward / inverse and the proposal kernel do not depend on it. */
int nnest_nvp_set_base(nnest_nvp_t *nvp, float beta);
/* netG.load_state_dict / state_dict (trainer.py:102-106, :241): host<->device copy of the packed
 * weights.  These two synchronise `stream` before returning. */
int nnest_nvp_load_weights(nnest_nvp_t *nvp, const float *packed_host, void *stream);
int nnest_nvp_store_weights(nnest_nvp_t *nvp, float *packed_host, void *stream);
/* device pointer of the packed weights / Adam exp_avg / exp_avg_sq (read-only views for tests) */
int nnest_nvp_device_ptrs(nnest_nvp_t *nvp, float **w_dev, float **m_dev, float **v_dev);
/* host copies of Adam's exp_avg / exp_avg_sq (torch.optim.Adam state, trainer.py:121-122); synchronises `stream` */
int nnest_nvp_store_adam(nnest_nvp_t *nvp, float *exp_avg_host, float *exp_avg_sq_host, void *stream);
int nnest_nvp_load_adam(nnest_nvp_t *nvp, const float *exp_avg_host, const float *exp_avg_sq_host, void *stream);
/* Adam step counter and optimiser reset (torch.optim.Adam state, trainer.py:121-122) */
int nnest_nvp_adam_state(nnest_nvp_t *nvp, int *step_count, int set_step, int reset_moments, void *stream);

/* NormalizingFlow.forward (networks.py:24-32) via Trainer.forward (trainer.py:247-257): x -> z, logdet */
int nnest_nvp_forward(nnest_nvp_t *nvp, const float *x_dev, float *z_dev, float *logdet_dev, int N, void *stream);
/* NormalizingFlow.inverse (networks.py:34-42) via Trainer.inverse (trainer.py:259-269): z -> x, logdet */
int nnest_nvp_inverse(nnest_nvp_t *nvp, const float *z_dev, float *x_dev, float *logdet_dev, int N, void *stream);
/* NormalizingFlowModel.log_probs (networks.py:71-76), N(0,I) base (networks.py:51-57) */
int nnest_nvp_log_probs(nnest_nvp_t *nvp, const float *x_dev, float *logp_dev, int N, void *stream);

/* Fused "one eval": x = f^-1(z), logdet; box prior UniformPrior(-1,1) (priors.py:39-43);
 * logl = loglike(like->scale * x) as safe_loglike (sampler.py:110-133) incl. non-finite -> -1e100.
 * logl_dev is float64 [N]; inbox_dev int32 [N] (1 = inside the box).  x_dev/logdet_dev may be NULL. */
int nnest_nvp_inverse_loglike(nnest_nvp_t *nvp, const nnest_like_t *like, const float *z_dev, float *x_dev,
                              float *logdet_dev, double *logl_dev, int *inbox_dev, int N, void *stream);

/* Likelihood.__call__ over rows (likelihoods.py:14-22) through safe_loglike: logl[n] = loglike(like->scale*x[n]).
 * x_unit_dev float32 [N,D]; logl_dev float64 [N]. */
int nnest_loglike(const nnest_like_t *like, const float *x_unit_dev, double *logl_dev, int N, int D,
                  void *stream);

/* Sampler._mcmc_sample, hard-constraint branch (sampler.py:229-463), `steps` Metropolis steps for C
 * walkers inside ONE launch.
 *   z_dev [C,D]        in: latent start (= forward(init_samples), sampler.py:264); out: final latent
 *   x_dev [C,D]        out: final x = f^-1(z) (sampler.py:266, :439)
 *   logl_dev [C] f64   in: init_loglikes; out: final log-likelihoods
 *   loglstar           hard constraint logl > loglstar (sampler.py:361)
 *   step_size          initial proposal scale (sampler.py:255)
 *   noise_dz_dev       NULL -> in-kernel noise: every (walker, lane group) and every walker's uniform draw runs a
 *                      xoshiro128++ stream (add / xor / rotate: full-rate VALU) SEEDED by one Philox4x32-10 block keyed by
 *                      (seed; walker_offset + walker, lane group, stream), the normals by Box-Muller -- so a walker's draws
 *                      depend on (seed, its global index) only and a sharded batch draws what the unsharded one draws
 *                      (csrc/flow_tile.h "proposal stream of the persistent MH kernel"; nnest_mh_fill_noise replays them);
 *                      else recorded noise [steps, C, D] (torch.randn_like(z), sampler.py:310)
 *   noise_u_dev        recorded uniforms [steps, C] (torch.rand, sampler.py:334); required iff noise_dz_dev
 *   hist_x_dev         optional [C, steps+1, D] history of x (reference return layout, sampler.py:455);
 *   hist_logl_dev      optional [C, steps+1] f64
 *   n_accept_dev [C]   out int32: accepted moves per walker  (sampler.py:418-420) in bits 0..29; bit 30 (NNEST_MH_ALL_MOVED) is set
 *                      when EVERY coordinate of the chain's last x differs from its first x = f^-1(z_0) -- the reference's test
 *                      of a chain before its end may replace a live point (nested.py:432: np.all(samples[:,0] != samples[:,-1]));
 *                      evaluated against the first x the launch computed -- in-kernel by the solo and quad forms (the first x
 *                      parked in x_dev meanwhile), by a follow-up kernel on the launch's stream for the 16-walker-tile forms
 *                      and the spline / MAF flows (the first x in a side buffer of the library); x_dev NULL: the bit says
 *                      "accepted at least once"
 *   n_call_dev [C]     out int32: likelihood calls per walker (rows that passed the prior/Jacobian test,
 *                      sampler.py:358-363)
 *   scale_out_dev      optional float32 [ngroups]: final scale per adaptation group (sampler.py:422-431)
 *   sync_dev           NNEST_MH_DYNAMIC_BATCH only (else NULL): nnest_mh_sync_words(steps) 8-byte words, ZERO at the launch;
 *                      the last word is an error flag (non-zero: a bounded wait ran out).  Either the caller zeroes them in
 *                      front of every launch, or it keeps a double buffer of 2 x nnest_mh_sync_words(steps) words, zeroes it
 *                      once, passes the halves alternately and sets NNEST_MH_SYNC_ZERO_NEXT (the other half lies behind
 *                      sync_dev) or NNEST_MH_SYNC_ZERO_PREV (in front of it): the launch then zeroes the other half for the
 *                      next one (the solo form in-kernel, on spare waves: no fill launch in front of a K4 launch)
 */
int nnest_mh_constrained_steps(nnest_nvp_t *nvp, const nnest_like_t *like, float *z_dev, float *x_dev,
                               double *logl_dev, double loglstar, float step_size, int steps, int C, int flags,
                               const float *noise_dz_dev, const float *noise_u_dev, uint64_t seed,
                               uint64_t walker_offset, float *hist_x_dev, double *hist_logl_dev, int *n_accept_dev,
                               int *n_call_dev, float *scale_out_dev, void *sync_dev, void *stream);
/* The kernel form (NNEST_MH_FORM_*, never AUTO) nnest_mh_constrained_steps runs for C walkers under `flags` (in-kernel noise, no
 * history): the pinned form of flags bits 16..19 if it applies to this flow shape, population and step rule, the form chosen
 * by population otherwise; -1 if the launch would be refused (NNEST_E_UNSUPPORTED).  A caller that shards one batch of C_total
 * walkers over ranks asks with C = C_total and pins the answer on every shard (reference: the MPI scatter of one batch,
 * nnest/nested.py:405-427, has no such choice -- every rank runs the same Python). */
int nnest_mh_form_for(const nnest_nvp_t *nvp, int C, int flags);

/* SLICE proposal in latent space (BASELINE.json north_star: "the slice/MH proposal step in latent space").
 * ABSENT FROM THE REFERENCE: Sampler._mcmc_sample proposes Gaussian random-walk Metropolis moves only (nnest/sampler.py:310-316), so
 * this step is build-defined and its parity UNPINNED; it is held to a CPU restatement of the same definition
 * (oracle/oracle.py::slice_sample).  `steps` slice-sampling updates (Neal 2003: stepping out, shrinkage) of every walker along a
 * fresh random direction of latent space, of the target the reference's constrained Metropolis step leaves invariant
 * (sampler.py:326-361): density |det dx/dz| on {x(z) in the unit box, logL(x(z)) > loglstar}.  Per step: eps ~ N(0, I),
 * candidates z + t * width * eps; log y = log|det|(z) + log u1; bracket [-u0, 1 - u0], stepped out by 1 up to max_stepout times per
 * side while the end lies in the slice; then up to max_shrink shrinkage draws t = t_l + (t_r - t_l) u_k.  z_dev [C,D] and logl_dev [C]
 * are updated in place, x_dev [C,D] receives the chains' ends; n_call_dev [C]: candidates whose likelihood decided (inside the box and
 * above the slice level), n_move_dev [C]: steps that moved, with NNEST_MH_ALL_MOVED as in nnest_mh_constrained_steps, n_eval_dev [C]
 * (or NULL): evaluations of the flow.  noise_dz_dev [steps,C,D] replays recorded directions (NULL: in-kernel Philox, the draws
 * nnest_slice_fill_noise exports); the uniforms are Philox4x32-10 words of (seed, walker, 64 step + k), exact in float32.
 * hist_x_dev [C, steps + 1, D] or NULL.  One walker per wave, no cross-workgroup wait: any C.  Shapes: the reference's defaults
 * (hidden 16, 3 blocks, 1 layer, scale ''), x_dim <= 128; NNEST_E_UNSUPPORTED otherwise.  (Added within ABI 15.) */
int nnest_slice_steps(nnest_nvp_t *nvp, const nnest_like_t *like, float *z_dev, float *x_dev, double *logl_dev, double loglstar,
                      float width, int steps, int C, int max_stepout, int max_shrink, const float *noise_dz_dev, uint64_t seed,
                      uint64_t walker_offset, float *hist_x_dev, int *n_call_dev, int *n_move_dev, int *n_eval_dev, void *stream);
int nnest_slice_fill_noise(float *dz_dev, int steps, int C, int D, uint64_t seed, uint64_t walker_offset, void *stream);
/* size of sync_dev in 8-byte words for a launch of `steps` steps */
int nnest_mh_sync_words(int steps);
/* number of adaptation groups nnest_mh_constrained_steps uses for C walkers (size of scale_out_dev) */
int nnest_mh_num_groups(const nnest_nvp_t *nvp, int C);

/* The in-kernel proposal noise as arrays, for tests: dz[steps,C,D] ~ N(0,1), u[steps,C] ~ U[0,1),
 * bit-identical to what nnest_mh_constrained_steps draws for (seed, walker_offset). */
int nnest_mh_fill_noise(float *dz_dev, float *u_dev, int steps, int C, int D, uint64_t seed, uint64_t walker_offset,
                        void *stream);

/* Trainer.train's epoch loop (trainer.py:198-232) inside one launch: for each epoch, Trainer._train
 * (trainer.py:384-403: minibatches of `batch` rows in the order perm_dev[epoch], data + jitter*noise,
 * loss = -mean(log_probs), backward, Adam with coupled weight decay) then Trainer._validate
 * (trainer.py:405-418), early stopping with `patience` and best-model restore (trainer.py:205-209, :241).
 *   xtrain_dev [n_train,D], xvalid_dev [n_valid,D] float32
 *   perm_dev [max_epochs, n_train] int32 (DataLoader shuffle order per epoch, trainer.py:185)
 *   noise_dev  NULL -> in-kernel normals, one Philox4x32-10 block + Box-Muller per (seed; position of the row in the epoch's
 *              order, epoch_offset + epoch, group of four dims) -- no stream state: any kernel form draws the same jitter;
 *              else [max_epochs, n_train, D]
 *              in perm order (torch.randn_like(data), trainer.py:392)
 *   losses_dev optional float32 [max_epochs, 2]: (train, validation) loss per epoch, normalised as the
 *              reference logs them (/len(dataset), trainer.py:403, :418)
 *   result_dev see nnest_train_result_t (device memory; in/out when NNEST_TRAIN_RESUME is set)
 *   epoch_offset, flags: a long train() may be issued as several launches ("chunks") so that the shuffle
 *              table stays small: chunk k passes epoch_offset = epochs already run and NNEST_TRAIN_RESUME,
 *              which continues the early-stopping state held in result_dev and the best-weights snapshot;
 *              NNEST_TRAIN_FINALIZE (last chunk) restores the best-validation weights (trainer.py:241); that
 *              restore also happens whenever patience runs out.  A single-launch train() passes
 *              epoch_offset 0 and flags = NNEST_TRAIN_FINALIZE.
 * Adam moments and step count persist in the handle across calls (optimizer is created once,
 * trainer.py:121-122).
 */
typedef struct {
    int epochs_run;             /* total epochs run so far (including previous chunks) */
    int best_epoch;             /* 1-based epoch of the best validation loss (trainer.py:206) */
    float best_validation_loss;
    float last_train_loss;
    int counter;                /* epochs since the last improvement (trainer.py:209, :223) */
    int stopped;                /* 1 when counter > patience ended the run (trainer.py:225); 2: internal error (a bounded grid
                                 * barrier of the multi-CU kernel ran out: results invalid) */
} nnest_train_result_t;

enum { NNEST_TRAIN_RESUME = 1, NNEST_TRAIN_FINALIZE = 2,
       NNEST_TRAIN_ONE_CU = 4 /* keep the epoch loop on ONE workgroup (the round-1 kernel) instead of one workgroup per 16-row tile
                               * of the minibatch on eight compute units; both produce the same bits (A/B and test switch) */ };

int nnest_nvp_train(nnest_nvp_t *nvp, const float *xtrain_dev, int n_train, const float *xvalid_dev, int n_valid,
                    const int *perm_dev, const float *noise_dev, uint64_t seed, float jitter, int batch,
                    int max_epochs, int patience, float lr, float weight_decay, int epoch_offset, int flags,
                    float *losses_dev, nnest_train_result_t *result_dev, void *stream);

/* One minibatch: loss and dloss/dw (before weight decay) into grad_dev [num_params], no update.
 * For tests (reference: loss.backward(), trainer.py:400). x_dev [M,D]. loss_dev float32[1]. */
int nnest_nvp_loss_grad(nnest_nvp_t *nvp, const float *x_dev, int M, float *grad_dev, float *loss_dev, void *stream);

/* The flow as one stage of a composite model (FastSlowNormalizingFlowModel, networks.py:86-150): vector-Jacobian product
 * of one batch (M <= 128 rows).  With L = sum_rows gz . f(x) + gld * sum_rows logdet(x):  grad_dev = dL/dw (packed order),
 * gx_dev = dL/dx [M,D]. */
int nnest_nvp_vjp(nnest_nvp_t *nvp, const float *x_dev, const float *gz_dev, float gld, int M, float *grad_dev, float *gx_dev,
                  void *stream);
/* one torch.optim.Adam step (coupled weight decay, trainer.py:121-122) from a gradient computed outside nnest_nvp_train;
 * uses and advances the handle's Adam state (the step counter lives on the device: asynchronous on `stream`, like the passes) */
int nnest_nvp_adam_step(nnest_nvp_t *nvp, const float *grad_dev, float lr, float weight_decay, void *stream);
/* MAF handles (nnest_maf_create): one epoch of Trainer._train (trainer.py:384-403) queued by ONE call -- per minibatch of `batch`
 * (<= 128) consecutive rows of rows_dev [n_train, D] (the caller has applied the epoch's permutation and jitter): loss + gradient,
 * one Adam step (coupled weight decay), image rebuild; *loss_sum_dev += the minibatch's loss (the reference's running sum of
 * loss.item(), trainer.py:402).  Asynchronous on `stream`; nothing is read back. */
int nnest_maf_train_epoch(nnest_nvp_t *nvp, const float *rows_dev, int n_train, int batch, float lr, float weight_decay,
                          float *loss_sum_dev, void *stream);

/* training jitter when jitter < 0 (trainer.py:168-171): 0.2 * mean of the 2-nearest-neighbour distance
 * table (self distance 0 included) of samples_dev [N,D] float64; result to out_dev float64[1]. */
int nnest_training_jitter(const double *samples_dev, int N, int D, double *out_dev, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Neural-spline flow: SingleSpeedSpline(num_inputs=D, hidden_dim=H, num_blocks=B, num_bins=K, tail_bound)
 * (networks.py:708-715) = [ActNorm (:661-705), Invertible1x1Conv (:625-658), NSF_CL (:559-622)] x B.
 * Packed weights = the concatenated state_dict: per block  s[D] t[D] | L[D,D] S[D] U[D,D] |
 *   f1.net.{0,2,4,6}.{weight,bias} | f2.net.{0,2,4,6}.{weight,bias}.
 * perm = the B fixed permutation matrices P [B,D,D] of the 1x1 convolutions (networks.py:634-635: a plain attribute
 * of the reference module, NOT in its state_dict); identity until loaded.
 * Same calling conventions as the nnest_nvp_* functions they parallel.  num_bins must be 8 (the reference's value).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct nnest_spline nnest_spline_t;
int nnest_spline_create(int D, int H, int B, int K, float tail_bound, nnest_spline_t **out);
int nnest_spline_destroy(nnest_spline_t *spl);
int nnest_spline_num_params(const nnest_spline_t *spl);
int nnest_spline_set_base(nnest_spline_t *spl, float beta);   /* as nnest_nvp_set_base */
int nnest_spline_load_weights(nnest_spline_t *spl, const float *packed_host, const float *perm_host, void *stream);
int nnest_spline_store_weights(nnest_spline_t *spl, float *packed_host, float *perm_host, void *stream);
/* NormalizingFlow.forward / .inverse (networks.py:24-42), NormalizingFlowModel.log_probs (networks.py:71-76) */
int nnest_spline_forward(nnest_spline_t *spl, const float *x_dev, float *z_dev, float *logdet_dev, int N, void *stream);
int nnest_spline_inverse(nnest_spline_t *spl, const float *z_dev, float *x_dev, float *logdet_dev, int N, void *stream);
int nnest_spline_log_probs(nnest_spline_t *spl, const float *x_dev, float *logp_dev, int N, void *stream);
int nnest_spline_inverse_loglike(nnest_spline_t *spl, const nnest_like_t *like, const float *z_dev, float *x_dev,
                                 float *logdet_dev, double *logl_dev, int *inbox_dev, int N, void *stream);
/* Sampler._mcmc_sample (sampler.py:229-463) with the spline inverse: arguments as nnest_mh_constrained_steps.  Kernel forms (chosen
 * by the library; the walkers' results do not depend on the form beyond float32 rounding): one wave per 16 walkers at large
 * populations; four waves per 16-walker tile while the tiles fit two per CU; four waves per EIGHT walkers (each held in both
 * halves of the matrix-core columns: one spline evaluation per lane serves two groups of four dimensions) at x_dim > 32 under a
 * fixed step or NNEST_MH_DYNAMIC_BATCH while those tiles fit one per CU.  scale_out_dev has one entry per 16 walkers in every form. */
int nnest_spline_mh_constrained_steps(nnest_spline_t *spl, const nnest_like_t *like, float *z_dev, float *x_dev,
                                      double *logl_dev, double loglstar, float step_size, int steps, int C, int flags,
                                      const float *noise_dz_dev, const float *noise_u_dev, uint64_t seed,
                                      uint64_t walker_offset, float *hist_x_dev, double *hist_logl_dev, int *n_accept_dev,
                                      int *n_call_dev, float *scale_out_dev, void *sync_dev, void *stream);

/* The form of the proposal kernel nnest_spline_mh_constrained_steps runs for C walkers under `flags` (the three above), -1 if the
 * launch would be refused (the batch-wide rule on a grid that is not resident).  (Added within ABI 15.) */
enum { NNEST_SPLINE_MH_WAVE = 0, NNEST_SPLINE_MH_TEAM = 1, NNEST_SPLINE_MH_PAIR = 2 };
int nnest_spline_mh_form_for(const nnest_spline_t *spl, int C, int flags);

/* Training.  ActNorm's data-dependent initialisation (networks.py:698-705): s = -log std(x) (unbiased), t = -mean(x e^s)
 * block after block from the batch x_dev [N,D] -- in the reference this happens inside the first forward pass of a
 * fresh model, which under Trainer.train is the first (jittered) minibatch. */
int nnest_spline_actnorm_init(nnest_spline_t *spl, const float *x_dev, int N, void *stream);
/* loss = -mean(log_probs(x)) and its gradient wrt the packed weights (loss.backward(), trainer.py:394-400) */
int nnest_spline_loss_grad(nnest_spline_t *spl, const float *x_dev, int M, float *grad_dev, float *loss_dev, void *stream);
/* Trainer.train's epoch loop (trainer.py:198-241) as nnest_nvp_train, except that the loop is driven from the host
 * (two launches per minibatch, the epoch's books kept on the device): losses_host [max_epochs,2] and result_host are HOST
 * pointers, the best-validation weights are restored on return and the call synchronises `stream`.  The Adam moments
 * persist across calls like torch.optim.Adam's state.
 * nnest_spline_train_form (added within ABI 15): the form a minibatch of `batch` rows runs in -- 1: one row per workgroup, the
 * evaluation on eight lanes per item, weight gradients contracted over the rows (nnest_spline_rows.hip: hidden_dim 16, x_dim <= 64,
 * batch <= 128); 0: sixteen / eight rows per workgroup of four waves (nnest_spline_train.hip).  Both follow the same reference
 * arithmetic; they agree to rounding, not to the bit.  NNEST_SPL_ROWS=0 in the environment pins form 0. */
int nnest_spline_train_form(const nnest_spline_t *spl, int batch);
/* the spline flow as one stage of a composite model (FastSlowSpline, networks.py:718-731): as nnest_nvp_vjp / nnest_nvp_adam_step */
int nnest_spline_vjp(nnest_spline_t *spl, const float *x_dev, const float *gz_dev, float gld, int M, float *grad_dev, float *gx_dev,
                     void *stream);
int nnest_spline_adam_step(nnest_spline_t *spl, const float *grad_dev, float lr, float weight_decay, void *stream);
int nnest_spline_train(nnest_spline_t *spl, const float *xtrain_dev, int n_train, const float *xvalid_dev, int n_valid,
                       const int *perm_dev, const float *noise_dev, uint64_t seed, float jitter, int batch, int max_epochs,
                       int patience, float lr, float weight_decay, float *losses_host, nnest_train_result_t *result_host,
                       void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 'choleksy' flow: SingleSpeedCholeksy(num_inputs=D) (networks.py:162-239): y = L x + b, L lower triangular with
 * diag = softplus(unconstrained_diag) + 1e-3.  Packed weights = state_dict order: bias[D], lower_entries[D(D-1)/2]
 * (np.tril_indices(D, -1) order), unconstrained_diag[D].  Conventions as the nnest_nvp_* functions.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct nnest_chol nnest_chol_t;
int nnest_chol_create(int D, nnest_chol_t **out);
int nnest_chol_destroy(nnest_chol_t *chol);
int nnest_chol_num_params(const nnest_chol_t *chol);
int nnest_chol_set_base(nnest_chol_t *chol, float beta);
int nnest_chol_load_weights(nnest_chol_t *chol, const float *packed_host, void *stream);
int nnest_chol_store_weights(nnest_chol_t *chol, float *packed_host, void *stream);
int nnest_chol_forward(nnest_chol_t *chol, const float *x_dev, float *z_dev, float *logdet_dev, int N, void *stream);
int nnest_chol_inverse(nnest_chol_t *chol, const float *z_dev, float *x_dev, float *logdet_dev, int N, void *stream);
int nnest_chol_log_probs(nnest_chol_t *chol, const float *x_dev, float *logp_dev, int N, void *stream);
int nnest_chol_loss_grad(nnest_chol_t *chol, const float *x_dev, int M, float *grad_dev, float *loss_dev, void *stream);
int nnest_chol_adam_step(nnest_chol_t *chol, const float *grad_dev, float lr, float weight_decay, void *stream);

/* Host-side (no GPU involved): the rows of a chain file in the reference's text format -- Sampler._save_samples,
 * nnest/sampler.py:494-511: np.savetxt(fmt='%.5E'): '%.5E' numbers separated by one space, '\n' after every row -- into out_host
 * (capacity >= 14 * n_rows * n_cols + 1 bytes); formatted on `threads` host threads.  Returns the number of bytes written, -1 on
 * a bad argument. */
long nnest_format_rows_e5(const double *rows_host, long n_rows, int n_cols, char *out_host, long out_cap, int threads);

/* Host-side (no GPU involved): the rows "<tag>,<step>,<value>\n" of the scalar log (utils.ScalarWriter, which stands where the
 * reference keeps trainer.writer.add_scalar, nnest/nested.py:467) for a run of steps, the value written as Python's repr(float)
 * writes it (the shortest digits that read back to the same double, ".0" after an integral value, exponent form below 1e-4 and
 * from 1e16, "nan" / "inf").  out_host: capacity >= n * (strlen(tag) + 48) + 1 bytes.  Returns the number of bytes written, -1 on
 * a bad argument.  (Added within ABI 15: host-only, nothing else changed.) */
long nnest_format_scalar_rows(const char *tag, const long long *steps_host, const double *values_host, long n, char *out_host,
                              long out_cap);

/* Host-side (no GPU involved): the per-iteration body of the nested-sampling loop while the MCMC strategy is in force --
 * NestedSampler.run, nnest/nested.py:269-293 (worst live point, evidence update, dead-point append), :429-437 (consume the next
 * usable chain of the batch), :458-471 (volume shell, remaining-evidence test) -- as ONE call per event instead of ~7 us of
 * interpreter per iteration (2e5 iterations in a BASELINE config-2 run).  The loop returns to the caller whenever the reference
 * does something that is not this arithmetic, with the reason:
 *   NNEST_HOST_FINISHED      fraction_remain <= dlogz or it > max_iters (nested.py:269)
 *   NNEST_HOST_RETRAIN       first pass, or it % update_interval == 0 at the top of a pass (nested.py:311-314): train, then call
 *                            again with resume = NNEST_HOST_AFTER_TRAIN
 *   NNEST_HOST_NEED_SAMPLES  the batch is used up (nested.py:399): run the chains from loglstar (state.loglstar), hand over the new
 *                            endpoints, nb = 0, resume = NNEST_HOST_AFTER_SAMPLES
 *   NNEST_HOST_LOG           a point was accepted at it > 0, it % log_interval == 0 (nested.py:439-456, before `it` advances):
 *                            log, then resume = NNEST_HOST_AFTER_LOG
 *   NNEST_HOST_CHECKPOINT    `it` has just advanced to a multiple of log_interval (nested.py:473-485); resume = NNEST_HOST_TOP
 *   NNEST_HOST_DEAD_FULL     the dead-point buffers are full: grow them, resume = NNEST_HOST_TOP (nothing was changed)
 * Arithmetic: float64, the reference's operations in the reference's order; log Z by logaddexp as numpy computes it (libm exp /
 * log1p).  The information H is NOT updated here (numpy's vectorised exp is not libm's): per dead point the loop records log Z
 * before the update (dead_logz_prev), the caller forms the two exponentials with numpy and nnest_host_h_update runs the recurrence.
 * Dead point k: row dead_v[k] = [v (D) | derived (nd)], dead_logl[k], dead_logwt[k], dead_logz_prev[k]. */
enum { NNEST_HOST_FINISHED = 0, NNEST_HOST_RETRAIN = 1, NNEST_HOST_NEED_SAMPLES = 2, NNEST_HOST_LOG = 3, NNEST_HOST_CHECKPOINT = 4,
       NNEST_HOST_DEAD_FULL = 5 };
enum { NNEST_HOST_TOP = 0, NNEST_HOST_AFTER_TRAIN = 1, NNEST_HOST_AFTER_SAMPLES = 2, NNEST_HOST_AFTER_LOG = 3 };
typedef struct {
    double logz, logvol, fraction_remain, max_logl, loglstar;
    long long it, n_dead;
    int accept_point, nb, first_time, resume, worst, pad_;
} nnest_host_state_t;
int nnest_host_mcmc_consume(nnest_host_state_t *state, int N, int D, int nd, double *active_u, double *active_v, double *active_logl,
                            double *active_derived, const double *end_u, const double *end_v, const double *end_logl,
                            const unsigned char *moved, const double *end_derived, int C, double *dead_v, double *dead_logl,
                            double *dead_logwt, double *dead_logz_prev, long long dead_cap, double dlogz, long long max_iters,
                            long long update_interval, long long log_interval);
/* The same loop while 'rejection_prior' is the strategy in force (nnest/nested.py:322-334, :362-373 over
 * Sampler._rejection_prior_sample, nnest/sampler.py:529-543): the reference draws one prior sample per likelihood call until one lies
 * above loglstar; the draws are independent, so the caller evaluates them a block per launch and hands over the block's candidates
 * -- the indices (ascending) that were above the threshold when the block was made, cand_logl32 (the kernel's likelihood of
 * float32(x)) and cand_logl64 (the reference's float64 one) per candidate, their rows cand_u / transformed rows cand_v [n_cand, D],
 * derived [n_cand, nd].  Candidates are examined in order, each once; an iteration's call count is the number examined up to and
 * including the accepted one (blocks used up without a hit carry over in pending_calls).  Returns as nnest_host_mcmc_consume, plus
 *   NNEST_HOST_NEED_SAMPLES  no block / block used up: make one of prior.block_next draws, set prior.n / n_cand / pos = k = hits = 0,
 *                            resume = NNEST_HOST_AFTER_SAMPLES
 *   NNEST_HOST_EXPIRED       the strategy expired in the pass before (volume_switch, or the last 20 call counts average more than
 *                            mcmc_steps and MCMC is available: nested.py:328-334); nothing of the next pass has been done.
 * NNEST_HOST_LOG here is nested.py:374-378 ((it + 1) % log_interval == 0).  (Added within ABI 15: host-only.) */
enum { NNEST_HOST_EXPIRED = 6 };
typedef struct {
    long long pos, k, hits, n, n_cand;   /* the walk through the current block */
    long long pending_calls;             /* candidates examined since the last accepted one, in blocks used up */
    long long total_calls;               /* Sampler.total_calls (sampler.py:119) */
    long long block_next;                /* size of the next block */
    double ncs[20];                      /* the last 20 call counts (nested.py:325-326) */
    double mean_calls;
    int ncs_len, expired;
} nnest_host_prior_t;
int nnest_host_prior_consume(nnest_host_state_t *state, nnest_host_prior_t *prior, int N, int D, int nd, double *active_u,
                             double *active_v, double *active_logl, double *active_derived, const long long *cand_idx,
                             const double *cand_logl32, const double *cand_logl64, const double *cand_u, const double *cand_v,
                             const double *cand_derived, double *dead_v, double *dead_logl, double *dead_logwt,
                             double *dead_logz_prev, long long dead_cap, double dlogz, long long max_iters, long long log_interval,
                             double volume_switch, double mcmc_steps, int mcmc_valid);
/* h <- exp(logwt - total) * logl + exp(logz_prev - total) * (h + logz_prev) - total  over n dead points (nested.py:281-283), the two
 * exponentials supplied by the caller (e1, e2); every operation rounded by itself.  Returns the new h. */
double nnest_host_h_update(double h, const double *e1, const double *e2, const double *logl, const double *logz_prev,
                           const double *total, long long n);

#ifdef __cplusplus
}
#endif
#endif /* NNEST_HIP_H */
