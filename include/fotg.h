/*
 * fotg.h -- C-ABI of the MI355X-native Dense-Inverse-Search optical-flow engine (libfotg.so).
 *
 * This is the drop-in boundary for the reference's flow path: the C++ classes of the reference's
 * CUDA build (src/oflow.h:22-46 OFClass, src/patchgrid.h:13-86 PatGridClass,
 * src/refine_variational.h:35-57 VarRefClass, src/params.h:9-65 opt_params/img_params) are thin
 * header-only wrappers over these entry points (include/fotg/oflow.h, include/fotg/patchgrid.h).
 * Plain pointers and sizes only; no C++/torch types.  All image/flow pointers are DEVICE pointers
 * (hipMalloc / any HIP-visible allocation) unless the name says host.  Every call returns a status
 * code (0 = ok); the library never calls exit() -- the C++ shim reproduces the reference's
 * print-and-exit behaviour (src/common/cuda_helper.h:286-299).
 *
 * Numerics follow the reference's kroeger/ CPU implementation (the parity oracle), not the
 * reference's CUDA port, which deviates from it.
 */
#ifndef FOTG_H
#define FOTG_H

#ifdef __cplusplus
extern "C" {
#endif

#define FOTG_OK               0
#define FOTG_ERR_ARG          1   /* bad argument / unsupported parameter combination */
#define FOTG_ERR_HIP          2   /* a HIP runtime call failed (fotg_last_hip_error()) */
#define FOTG_ERR_BATCH        3   /* n > max_batch */
#define FOTG_ERR_UNSUPPORTED  4   /* valid in the reference but not implemented here: patch sizes other than 4 / 8 / 12 / 16; a coarsest
                                     level of fewer than 5 rows or 3 columns; with the refinement on: levels of more than 16 384 rows or whose
                                     skewed system array (w + h) x h x 32 bytes reaches 4 GB per pair (lexicographic solver; an 8K frame at
                                     full resolution is 1.7 GB; the depth mode and sor_coupled_slow_but_readable, FOTG_SOR_POINT, also run to 16 384 rows);
                                     red-black ordering in the depth mode */
#define FOTG_ERR_STALL        5   /* a bounded wait between workgroups of the tile solver (levels of more than 96 rows) timed out (preempted or
                                     starved producer) and the batch could not be recomputed.  The entry points that synchronise with the
                                     host and still have the call's inputs -- fotg_calc, fotg_pipe_wait(host_wait = 1), fotg_pipe_sync,
                                     fotg_node_wait -- recompute a stalled batch on the solver path without such waits and SUCCEED (the
                                     context's "stalls" counter counts them); they return this code only when that fails too, or for
                                     batches whose inputs are gone (fotg_pipe_wait(host_wait = 2), pulled pieces of fotg_node_submit_scatter).
                                     Callers of the asynchronous entry points ask fotg_ctx_counter(ctx, "take_stall") after their own
                                     synchronisation (it does not synchronise; read-and-clear) and re-submit. */

#define FOTG_SOR_LEXICOGRAPHIC 0  /* kroeger FDF1.0.1/solver.c:77-421 order (parity mode, default) */
#define FOTG_SOR_REDBLACK      1  /* red-black ordering of the same 2x2 block update (src/kernels/flowUtil.cu:297-362 ordering) */
#define FOTG_SOR_POINT         2  /* kroeger FDF1.0.1/solver.c:19-72 sor_coupled_slow_but_readable, the solver of the reference's OpenMP
                                     build (refine_variational.cpp:202-206), in the order of its serial loop: point update (du from the
                                     old dv, dv from the new du), no block inverse.  Compatibility mode (levels of <= 1024 rows). */

/* == optparam of kroeger/oflow.h:33-76 / opt_params of src/params.h:23-65 (explicitly set part) */
typedef struct fotg_params {
  int sc_f;            /* coarsest scale            (src: coarsest_scale) */
  int sc_l;            /* finest scale              (src: finest_scale) */
  int ps;              /* patch size 4, 8, 12 or 16 (src: patch_size; the operating points use 8 and 12) */
  int max_iter;        /* LK iterations             (src: grad_descent_iter) */
  int min_iter;
  float dp_thresh;     /* 0.05 (squared internally, kroeger/oflow.cpp:88; src/oflow.cpp:53) */
  float dr_thresh;     /* 0.95 */
  float res_thresh;    /* 0.0  */
  float patove;        /* patch overlap             (src: patch_stride) */
  int patnorm;         /* mean normalisation        (src: use_mean_normalization) */
  int noc;             /* channels: 1 gray (kroeger run_OF_INT), 3 interleaved (src/, run_OF_RGB) */
  int usetvref;        /* variational refinement    (src: use_var_ref) */
  float tv_alpha, tv_gamma, tv_delta;   /* 10, 10, 5 */
  int tv_innerit;      /* 1: inner iterations = tv_innerit*(level+1) */
  int tv_solverit;     /* 3                         (src: var_ref_iter) */
  float tv_sor;        /* 1.6                       (src: var_ref_sor_weight) */
  int sor_mode;        /* FOTG_SOR_* */
  int costfct;         /* patch cost: 0 L2 (all operating points), 1 L1, 2 pseudo-Huber (kroeger/oflow.h:45, patch.cpp:230-261) */
  float normoutlier;   /* 5.0: Huber threshold (kroeger/oflow.h:63; src: norm_outlier) */
  int usefbcon;        /* 0 (all operating points); 1: also compute the backward flow at every scale and merge both in the
                          densification (kroeger/oflow.h:44, oflow.cpp:160-170, patchgrid.cpp:278-375) */
  int depth;           /* 0: optical flow (the reference's SELECTMODE=1 build, run_OF_*); 1: stereo depth (SELECTMODE=2,
                          run_DE_*): ONE horizontal displacement per pixel -- every flow array (initflow, outflow, the
                          per-stage flow arguments, fotg_upsample_crop) has 1 channel instead of 2; forward grid /
                          refinement clamp the displacement to <= 0, the backward ones (usefbcon) to >= 0
                          (kroeger/oflow.cpp:76-80,153-157, patch.cpp:188-193, refine_variational.cpp:243-330) */
  int u8_color;        /* 0: 8-bit frames have `noc` channels.  1 / 2 (noc = 1 only): the 8-bit entry points (fotg_calc_batch_u8,
                          fotg_calc_sequence_u8, fotg_pipe_submit_u8, fotg_node_submit_u8) take THREE-channel frames (n x h_org x
                          w_org x 3 uint8), B,G,R byte order as cv::imread delivers (1) or R,G,B (2), and the flow is computed on
                          their gray value, converted on load with OpenCV's fixed-point formula (1868 B + 9617 G + 4899 R + 8192)
                          >> 14 -- what cv::imread(file, IMREAD_GRAYSCALE) feeds kroeger/run_dense.cpp:199-209 for a colour file.
                          Bit-identical to the gray 8-bit path on the converted frames; the float entry points still take gray. */
  int fast_math;       /* 0 (default): parity mode -- every kernel evaluates the reference's f32 expressions in the reference's order,
                          no fused multiply-add: results bit-identical to the CPU oracle.  1: tolerance mode -- the patch loop
                          (PatClass::OptimizeIter, kroeger/patch.cpp:159-212) runs an algebraically equivalent form with fused
                          multiply-adds, free reduction order and a precomputed inverse Hessian (csrc/lk_fast.hip.h): about a third
                          of the instructions, flows within the north star's 1e-3 px mean endpoint error of the parity mode (the
                          tests state the measured distances).  Applies to L2 cost, min_iter == max_iter, res_thresh <= 0, optical
                          flow -- every operating point; other configurations run the exact patch kernel regardless.  The
                          refinement's cell update (solvers of levels of more than 64 rows) uses fused multiply-adds and its data term
                          (compute_data / compute_smoothness, FDF1.0.1/opticalflow_aux.c:123-165,310-438) v_rcp / v_rsq instead of
                          the IEEE divisions and square roots (csrc/varref_dataterm.inc.h), optical flow only. */
} fotg_params;

typedef struct fotg_ctx fotg_ctx;

/* operating points 1..4: kroeger/run_dense.cpp:225-268 == src/run_dense.cpp:168-209 */
int fotg_op_point(int op, int width_org, int channels, fotg_params *out);
/* padding that makes W,H multiples of 2^sc_f: kroeger/run_dense.cpp:298-305 == src/run_dense.cpp:231-237 */
int fotg_padded_size(int w, int h, int sc_f, int *wp, int *hp, int *padw, int *padh);

/* Replaces OFClass::OFClass(opt_params, img_params) (src/oflow.cpp:38-145): allocates pyramids, per-scale
 * flow buffers, patch-grid state and refinement workspace for `max_batch` frame pairs of w_org x h_org
 * (unpadded).  Padding to multiples of 2^sc_f (src/run_dense.cpp:231-253, cu::pad) is folded into the
 * pyramid kernel, so callers pass the ORIGINAL frames; already padded frames work too (pad = 0). */
int fotg_create(const fotg_params *p, int w_org, int h_org, int device, int max_batch, fotg_ctx **out);
/* Replaces OFClass::~OFClass (src/oflow.cpp:147-179) */
void fotg_destroy(fotg_ctx *ctx);

/* Replaces OFClass::calc(I0, I1, iparams, initflow, outflow) (src/oflow.cpp:211-368) for n pairs at once.
 * I0, I1: n contiguous frames, each h_org x w_org x noc float32 interleaved (src/run_dense.cpp:137-162).
 * initflow: NULL (as every reference caller passes, src/run_dense.cpp:286) or n x (h/2^(sc_f+1)) x (w/2^(sc_f+1)) x 2
 *   (patches in the last row / column of an odd-sized coarsest level take the last row / column of it: the reference's
 *   InitializeFromCoarserOF indexes one past the array there).
 * outflow: n x (Hp/2^sc_l) x (Wp/2^sc_l) x 2 float32 interleaved (u,v), row-major (src/run_dense.cpp:280-289).
 * stream: hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing. */
int fotg_calc_batch(fotg_ctx *ctx, int n, const float *I0, const float *I1, const float *initflow,
                    float *outflow, void *stream);
/* The same for 8-bit frames (n x h_org x w_org x noc uint8, device memory): what cv::imread delivers before the
 * reference converts to float (src/run_dense.cpp:137-145).  The conversion is exact and happens on load in the pyramid
 * kernel, which then reads a quarter of the bytes.  Results are bit-identical to fotg_calc_batch on the converted frames. */
int fotg_calc_batch_u8(fotg_ctx *ctx, int n, const unsigned char *I0, const unsigned char *I1, const float *initflow,
                       float *outflow, void *stream);
/* Sequence mode (video): `frames` = n_frames consecutive frames (same layout as I0 above), outflow = the n_frames - 1
 * flows frame k -> frame k+1 (2 <= n_frames <= max_batch + 1).  Each frame's pyramid is built once and serves as the
 * target of pair k-1 and as the template source of pair k -- the reference rebuilds both pyramids for every pair
 * (kroeger/run_dense.cpp:331-336).  Bit-identical to fotg_calc_batch(frames[0..n-2], frames[1..n-1]). */
int fotg_calc_sequence(fotg_ctx *ctx, int n_frames, const float *frames, const float *initflow, float *outflow, void *stream);
int fotg_calc_sequence_u8(fotg_ctx *ctx, int n_frames, const unsigned char *frames, const float *initflow, float *outflow,
                          void *stream);
/* ---- batches in flight (no reference equivalent: the reference's calc() is synchronous, one pair at a time) -------------
 * A pipe owns `depth` engine contexts, each on an internal non-blocking stream.  fotg_pipe_submit enqueues one batch exactly
 * like fotg_calc_batch (same arguments, same bits) on the next context in turn and returns at once; up to `depth` batches
 * overlap on the GPU.  The work starts behind everything enqueued so far on `after_stream` (the stream that produced the
 * frames; NULL = default stream), or at once with after_stream = FOTG_NO_STREAM (frames already in place; note that an event
 * on a busy stream is only reached when that stream's queue has drained).  The caller keeps I0 / I1 / outflow alive and
 * untouched until the ticket has been waited for.  A pipe is used from one thread at a time (like a context); batches complete
 * in submission order per slot; every ticket has a completion event of its own for the next 4 * depth submissions (a wait for an
 * older ticket waits for a later batch of the same slot, which covers it). */
#define FOTG_PIPE_MAX_DEPTH 8
#define FOTG_NO_STREAM ((void *)(-1))
typedef struct fotg_pipe fotg_pipe;
int fotg_pipe_create(const fotg_params *p, int w_org, int h_org, int device, int max_batch, int depth, fotg_pipe **out);
void fotg_pipe_destroy(fotg_pipe *pipe);
int fotg_pipe_submit(fotg_pipe *pipe, int n, const float *I0, const float *I1, const float *initflow, float *outflow,
                     void *after_stream, long *ticket);
int fotg_pipe_submit_u8(fotg_pipe *pipe, int n, const unsigned char *I0, const unsigned char *I1, const float *initflow,
                        float *outflow, void *after_stream, long *ticket);
/* The same with the element type as an argument (u8 = 0: float32 frames, 1: 8-bit frames) and flags:
 * FOTG_SUBMIT_NO_RECOMPUTE  the batch's frames or outflow may be gone before the host waits for the ticket (staging buffers that are
 *                           recycled behind a device-side wait): a flagged stall is reported for it (FOTG_ERR_STALL), never recomputed. */
#define FOTG_SUBMIT_NO_RECOMPUTE 1
int fotg_pipe_submit_ex(fotg_pipe *pipe, int n, const void *I0, const void *I1, int u8, const float *initflow, float *outflow,
                        void *after_stream, int flags, long *ticket);
/* RECOMPUTE CONTRACT.  A host wait that finds a context's stall word set recomputes the unverified batches of that context from the
 * pointers of their submits.  That is legal only for tickets whose buffers are still in place, so a ticket is recomputed only if it was
 * submitted without FOTG_SUBMIT_NO_RECOMPUTE AND has not been handed out through fotg_pipe_wait(host_wait = 0) or
 * fotg_pipe_ticket_event: after such a hand-over the waiting stream owns the result and the caller may free or reuse I0 / I1 /
 * outflow as soon as its own wait is over.  Suspects that cannot be recomputed -- those, and tickets older than the 4 * depth
 * submissions the pipe keeps arguments for -- are reported: FOTG_ERR_STALL from every host wait for that ticket and from
 * fotg_pipe_sync (their flows are not valid; re-submit).
 * host_wait = 0: `stream` (NULL = default stream) waits for batch `ticket` on the device, the call returns at once;
 * host_wait = 1: the calling thread waits; if the context of the batch has flagged a timed-out inter-workgroup wait, the batches of
 *   that context that have not been verified yet are recomputed (from the arguments of their submits -- which the caller keeps in
 *   place until a ticket has been waited for) and the call succeeds;
 * host_wait = 2: the calling thread waits; a flagged batch is reported (FOTG_ERR_STALL, on every wait for that ticket) instead of
 *   recomputed -- for callers whose frames are not in place any more.
 * Per-ticket state is kept for the last 4 * depth submissions; a suspect older than that keeps FOTG_ERR_STALL (per-slot range).  May be
 * called from another thread than the one that submits. */
int fotg_pipe_wait(fotg_pipe *pipe, long ticket, void *stream, int host_wait);
/* the calling thread waits for everything submitted so far */
int fotg_pipe_sync(fotg_pipe *pipe);
/* the completion event (a hipEvent_t) of batch `ticket` (FOTG_ERR_ARG for a ticket that has not been submitted), for waiting on
 * several pipes at once: valid for the next 4 * depth submissions (afterwards it belongs to a later batch of the same slot).  Note
 * the head-of-line effect of that: with more than 4 * depth newer submissions outstanding, a wait for an old ticket waits for a
 * later batch of its slot (a latency cost, never a correctness one). */
int fotg_pipe_ticket_event(fotg_pipe *pipe, long ticket, void **event);
/* the engine context of a slot (geometry queries, taps, counters) */
int fotg_pipe_context(fotg_pipe *pipe, int slot, fotg_ctx **ctx);

/* ---- one process, several GPUs (no reference equivalent: src/run_dense.cpp:277-289 drives one device) ------------
 * Frame pairs are independent, so a batch of n pairs is cut into contiguous shards -- slot d gets the pairs [begin, begin + count)
 * with count = n / ndev (+ 1 for the first n % ndev slots) and begin = d * (n / ndev) + min(d, n % ndev); ALWAYS take them from
 * fotg_node_shard -- and every slot runs its shard through a pipe of its own (above) on its device,
 * issued by a host thread of its own; there is no exchange between the GPUs on the data path.  `devices` may name a device more
 * than once (two slots on one GPU).  max_batch = pairs per pipe submission on ONE device (a shard larger than that runs as
 * consecutive pieces on consecutive pipe slots), depth = batches in flight per device.  Up to 16 submitted jobs may be waiting to
 * be waited for (FOTG_ERR_BATCH beyond that).  Results are bit-identical to fotg_calc_batch on the same pairs. */
#define FOTG_NODE_MAX_DEV 16
typedef struct fotg_node fotg_node;
int fotg_node_create(const fotg_params *p, int w_org, int h_org, const int *devices, int ndev, int max_batch, int depth, fotg_node **out);
void fotg_node_destroy(fotg_node *node);
/* [begin, begin + count) of the pairs of slot d */
int fotg_node_shard(int n, int ndev, int d, int *begin, int *count);
/* resident frames: I0[d], I1[d], outflow[d] = the shard of slot d in the memory of devices[d] (frames / flows laid out as in
 * fotg_calc_batch; entries of empty shards are ignored).  Returns at once; the frames must be in place (the work starts
 * immediately) and, like outflow, stay untouched until fotg_node_wait(ticket). */
int fotg_node_submit(fotg_node *node, int n, const float *const *I0, const float *const *I1, float *const *outflow, long *ticket);
int fotg_node_submit_u8(fotg_node *node, int n, const unsigned char *const *I0, const unsigned char *const *I1, float *const *outflow, long *ticket);
/* scatter / gather: I0, I1 (n frames each) and outflow (n flows) live on devices[0].  Slot 0 computes its shard in place; every
 * other slot pulls its shard over xGMI in chunks of `chunk` <= max_batch pairs (hipMemcpyPeerAsync into depth + 1 staging buffers
 * on a copy stream of its own), computes chunk t while chunk t + 1 travels, and writes its flows back into `outflow`. */
int fotg_node_submit_scatter(fotg_node *node, int n, const float *I0, const float *I1, float *outflow, int chunk, long *ticket);
/* the same with 8-bit frames (layout as in fotg_calc_batch_u8, three bytes per pixel with u8_color): the shards travel as bytes */
int fotg_node_submit_scatter_u8(fotg_node *node, int n, const unsigned char *I0, const unsigned char *I1, float *outflow, int chunk, long *ticket);
/* The calling thread waits for job `ticket` and every job before it on all devices.  A piece whose tile solver gave up a bounded
 * wait is recomputed where its frames are still in place (resident shards, the source slot of a scatter: the call then succeeds);
 * pulled pieces of a scatter cannot be (their staging buffers have been recycled): FOTG_ERR_STALL, re-submit the job.  Returns the
 * worst status of the jobs this call covers; every job keeps its own status for later (repeated, out-of-order) waits for it, for
 * the next 16 jobs.  fotg_node_last_hip_error: the HIP error behind the last FOTG_ERR_HIP a wait returned (it was raised on a worker
 * thread, where fotg_last_hip_error() of the waiting thread does not see it). */
int fotg_node_wait(fotg_node *node, long ticket);
int fotg_node_last_hip_error(const fotg_node *node);
int fotg_node_sync(fotg_node *node);
int fotg_node_info(const fotg_node *node, int *ndev, int *out_w, int *out_h, int *flow_channels);
int fotg_node_pipe(fotg_node *node, int slot, fotg_pipe **pipe);

/* Single pair, outflow in HOST memory, synchronous -- the exact shape of the reference call. */
int fotg_calc(fotg_ctx *ctx, const float *I0, const float *I1, const float *initflow, float *outflow_host);

/* kroeger/run_dense.cpp:407-414 == src/run_dense.cpp:293-303: flow *= 2^sc_l, bilinear x2^sc_l, crop the
 * padding.  in: n x hl x wl x 2 (device), out: n x h_org x w_org x 2 (device). */
int fotg_upsample_crop(fotg_ctx *ctx, int n, const float *flow, float *out, void *stream);

/* Gradient-magnitude input, the reference's SELECTCHANNEL==2 build (kroeger/run_dense.cpp:138-147): level 0 of the pyramid is
 * sqrt(dx^2 + dy^2) of the padded frame (cv::Sobel ksize 1, REFLECT_101 at the padded edge).  frames: n x h_org x w_org x
 * channels (device, f32 or 8-bit); out: n x Hp x Wp x channels f32 (device), Wp / Hp = fotg_padded_size(w_org, h_org, sc_f) --
 * the replicate padding (run_dense.cpp:306-310) is part of the call, so the flow context for these frames is created with
 * w_org = Wp, h_org = Hp (no further padding) and fotg_upsample_crop of a context of the original size crops the result. */
int fotg_gradient_magnitude(int device, int n, const float *frames, int w_org, int h_org, int channels, int sc_f, float *out, void *stream);
int fotg_gradient_magnitude_u8(int device, int n, const unsigned char *frames, int w_org, int h_org, int channels, int sc_f, float *out, void *stream);

/* op.verbosity of the reference (src/oflow.cpp:246-365, kroeger/oflow.cpp:298-360).  0 (default): silent, asynchronous.
 * > 0: every flow call (fotg_calc, fotg_calc_batch, ...) waits for its launches and prints "TIME (O.Flow Run-Time   ) (ms): ..."
 * (the flow without the pyramid, like the reference); > 1: also one "TIME (Sc: .., #p: .., pconst, pinit, poptim, cflow, tvopt,
 * total): ..." line per scale, from HIP-event times of the stages on the launch stream (patch construction and initialisation
 * are part of the LK launch: pconst = pinit = 0). */
int fotg_set_verbosity(fotg_ctx *ctx, int verbosity);
/* the five times (ms) of `level` measured by the last flow call with verbosity > 0: pconst, pinit, poptim, cflow, tvopt
 * (PatGridClass::printTimings, src/patchgrid.cpp:334-345, prints from these) */
int fotg_level_timings(fotg_ctx *ctx, int level, float *ms5);

/* geometry queries */
int fotg_level_size(const fotg_ctx *ctx, int level, int *w, int *h);        /* unpadded level size */
int fotg_out_size(const fotg_ctx *ctx, int *w, int *h);                     /* finest-scale flow size */
int fotg_num_patches(const fotg_ctx *ctx, int level, int *nopw, int *noph); /* PatGridClass::GetNumPatches{W,H} */

/* ---- per-stage entry points (what PatGridClass / VarRefClass / the pyramid helpers bind to) ---- */

/* cu::constructImgPyramids (src/kernels/pyramid.cpp:32-223) with kroeger semantics (run_dense.cpp:130-178).
 * which: 0 -> I0 (image + gradients), 1 -> I1 (image only; its gradients are never read, patch.cpp:266). */
int fotg_pyramid(fotg_ctx *ctx, int n, const float *I, int which, void *stream);
/* both frames of n pairs in shared launches (what fotg_calc_batch does).  stages: bit 0 = the HBM-streaming base
 * kernel (frames -> level min(sc_l,4)), bit 1 = coarser levels + borders + gradients; 3 = everything. */
int fotg_pyramid_pair(fotg_ctx *ctx, int n, const float *I0, const float *I1, int stages, void *stream);
/* the same from 8-bit frames (noc channels, or three with fotg_params::u8_color): what the 8-bit flow entry points run first */
int fotg_pyramid_pair_u8(fotg_ctx *ctx, int n, const unsigned char *I0, const unsigned char *I1, int stages, void *stream);
/* device pointer of a pyramid plane of pair 0 (pairs are `*pair_stride` floats apart).
 * kind: 0 image, 1 dx, 2 dy.  Layout (h_l+2ps) x (w_l+2ps) x noc, like the reference's padded levels. */
int fotg_level_ptr(fotg_ctx *ctx, int which, int level, int kind, float **ptr, long *pair_stride);

/* PatGridClass (src/patchgrid.h:13-86).  Grid state lives in the context, one grid per level.
 * All level images use the padded layout above; n pairs, `pair_stride` floats apart. */
int fotg_grid_init(fotg_ctx *ctx, int level, int n, const float *I0, const float *I0x, const float *I0y,
                   long pair_stride, void *stream);                                   /* InitializeGrid */
int fotg_grid_set_target(fotg_ctx *ctx, int level, const float *I1, long pair_stride);  /* SetTargetImage */
int fotg_grid_init_from_coarser(fotg_ctx *ctx, int level, int n, const float *flow_prev, void *stream); /* InitializeFromCoarserOF */
/* depth mode only: camera side of the level's grid and of fotg_varref on that level, camparam::camlr (kroeger/oflow.h:28):
 * 0 = left camera, displacement <= 0 (default; what the forward grid uses), 1 = right camera, displacement >= 0 */
int fotg_grid_set_camera(fotg_ctx *ctx, int level, int camlr);
int fotg_grid_optimize(fotg_ctx *ctx, int level, int n, void *stream);                /* Optimize */
int fotg_grid_aggregate(fotg_ctx *ctx, int level, int n, float *flowout, void *stream); /* AggregateFlowDense */
/* test taps: copy grid state of pair `pair` to host.  Any pointer may be NULL.
 * p_iter: nop x 2, pweight: nop x nv, tmpl/tdx/tdy: nop x nv, hes: nop x 3, cnt: nop ints */
int fotg_grid_read(fotg_ctx *ctx, int level, int pair, float *p_iter, float *pweight, float *tmpl,
                   float *tdx, float *tdy, float *hes, int *cnt);
/* allocate the optional tap buffers (templates, Hessians, iteration counts) that fotg_grid_read returns;
 * off by default so the production path writes nothing it does not need */
int fotg_enable_taps(fotg_ctx *ctx, int on);
/* per-iteration trace for tests: host buffer nop x (max_iter+1) x 4 [p0,p1,mares,cnt] of pair 0; the next
 * fotg_grid_optimize on this level fills it (synchronously).  NULL disables. */
int fotg_grid_set_trace(fotg_ctx *ctx, int level, float *trace_host);

/* VarRefClass::VarRefClass(I0, I1, iparams, op, flowout) (src/refine_variational.cpp:31-145):
 * refines `flow` (n x h_l x w_l x 2, device) in place. */
int fotg_varref(fotg_ctx *ctx, int level, int n, const float *I0, const float *I1, long pair_stride,
                float *flow, void *stream);
/* test tap: copy one refinement workspace plane (stride-padded, FDF image_t layout) of pair `pair` to host.
 * The solver planes (du, dv, a11 .. sv) of levels refined entirely on chip are only written back when
 * fotg_enable_taps(ctx, 1) was called before fotg_varref.
 * name: "wx","wy","mask","du","dv","sh","sv","a11","a12","a22" (block inverse),"b1","b2","avg","Iz","Ix","Iy","Ixx","Ixy","Iyy","Ixz","Iyz"
 * depth mode: "wx","mask","du","uu","s","a11","b1","sh","sv" and the image planes (a11/b1: the scalar system of compute_data_DE) */
int fotg_varref_plane(fotg_ctx *ctx, int pair, const char *name, int level, float *host_out);

/* measurement tap: ONE sor_coupled call (the launch the refinement issues once per inner iteration) of `level` for n pairs on the
 * system the last fotg_varref left in the workspace; bench.py times it for the roofline of the time-dominant kernel */
int fotg_bench_sor_call(fotg_ctx *ctx, int level, int n, void *stream);
/* test tap: how often a kernel variant was launched by this process ("sor_stream", "sor_tiles"); -1 for unknown names */
long fotg_debug_counter(const char *name);
/* per-context counters.  "take_stall" (does NOT synchronise; the query for callers of the asynchronous entry points, AFTER their own
 * synchronisation): 1 = a bounded inter-workgroup wait of this context timed out since the last query / the last FOTG_ERR_STALL
 * -- the flows computed since then are not valid --, 0 = none; it clears the flag.  "stalls" (does NOT synchronise, does not
 * clear): how many such time-outs the host has seen so far (+ 1 while one is pending).  "inject_stall" raises the flag as a timed-out
 * wait would -- only in contexts created with FOTG_TEST_TAPS=1 in the environment (tests), -1 otherwise.
 * Test tap that synchronises the device: "tile_timeouts" = device-side count of those time-outs of the tile solver since the
 * context was created -- 0 unless something is broken; -1 for unknown names */
long fotg_ctx_counter(fotg_ctx *ctx, const char *name);
const char *fotg_strerror(int status);
int fotg_last_hip_error(void);
const char *fotg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FOTG_H */
