/*
 * dis_oracle.h -- CPU ORACLE for the DIS optical-flow hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm in the reference's kroeger/ CPU implementation
 * (OF_DIS v1.0.1).  It is the checker the HIP engine is compared against; it is never the thing
 * measured or shipped.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product library (flowonthego_amd/csrc -> libfotg.so) does not link or call it.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - the FDF1.0.1 stages (warp, derivatives, smoothness, data term, sub_laplacian, sor_coupled)
 *     are checked bit-for-bit against the reference's own C sources compiled unmodified into
 *     oracle/_ref/libfdf_ref_{gray,rgb}.so (oracle/Makefile, target `ref`);
 *   - the whole pipeline is checked against the reference's only golden result,
 *     kroeger/flows/alley_0001.flo (tests/golden/), mean EPE <= 0.05 px;
 *   - the C++ stages (patch.cpp, patchgrid.cpp, oflow.cpp) need Eigen, which this image lacks, so
 *     they cannot be built here; the OpenCV calls of run_dense.cpp likewise.  Their restatement
 *     is pinned only through the golden .flo.  Summation order of the per-patch reductions is
 *     Eigen-version dependent in the reference; the order used here is documented at dis_sum().
 */
#ifndef DIS_ORACLE_H
#define DIS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* optparam of kroeger/oflow.h:33-76, explicitly set part, plus noc (SELECTCHANNEL 1 -> 1, 3 -> 3) */
typedef struct dis_params {
  int sc_f;          /* first (coarsest) scale */
  int sc_l;          /* last (finest) scale */
  int ps;            /* p_samp_s, patch edge length */
  int max_iter;
  int min_iter;
  float dp_thresh;   /* NOT squared; squared internally like oflow.cpp:88 */
  float dr_thresh;
  float res_thresh;
  float patove;      /* patch overlap 0..1 */
  int patnorm;
  int noc;           /* 1 gray, 3 RGB */
  int usetvref;
  float tv_alpha, tv_gamma, tv_delta;
  int tv_innerit, tv_solverit;
  float tv_sor;
  int costfct;       /* oflow.h:45: 0 L2, 1 L1, 2 pseudo-Huber (patch.cpp:230-261); the operating points use 0 */
  float normoutlier; /* oflow.h:63: 5.0, Huber threshold b */
  int usefbcon;      /* oflow.h:44: merge forward and backward flow (patchgrid.cpp:278-375); the operating points use 0 */
  int depth;         /* 0: optical flow (SELECTMODE 1); 1: stereo depth (SELECTMODE 2): one horizontal displacement per
                        pixel/patch (oflow.cpp:76-80 nop = 1), flow arrays have ONE channel */
} dis_params;

/* kroeger/run_dense.cpp:180-183 and :225-268.  op in 1..4 (anything else -> 2). */
int dis_auto_first_scale(int imgwidth, int fratio, int patchsize);
void dis_op_point(int op, int width_org, int noc, dis_params *p);

/* kroeger/run_dense.cpp:298-311: replicate-pad so W,H are multiples of 2^sc_f.
 * Returns padded sizes; out must hold Wp*Hp*noc floats.  padw/padh returned. */
void dis_padded_size(int w, int h, int sc_f, int *wp, int *hp, int *padw, int *padh);
void dis_pad_frame(const float *in, int w, int h, int noc, int sc_f, float *out);
/* kroeger/run_dense.cpp:138-147 (SELECTCHANNEL==2): gradient magnitude of a (padded) frame as the pyramid's input */
void dis_gradient_magnitude(const float *in, int w, int h, int noc, float *out);

/* kroeger/run_dense.cpp:130-178.  img: Wp x Hp x noc interleaved f32.  For every level l in
 * 0..sc_f allocates (malloc) padded image / dx / dy of size (w_l+2ps) x (h_l+2ps) x noc.
 * Free with dis_pyramid_free. */
typedef struct dis_pyramid {
  int nlev, noc, ps, w0, h0;
  float **im, **dx, **dy;
} dis_pyramid;
dis_pyramid *dis_pyramid_build(const float *img, int wp, int hp, int noc, int sc_f, int ps);
void dis_pyramid_free(dis_pyramid *p);
int dis_level_w(const dis_pyramid *p, int l);
int dis_level_h(const dis_pyramid *p, int l);

/* The reduction every per-patch sum uses (template mean, Hessian, projections, query mean,
 * L1 residual).  Element e of n belongs to pixel e/noc; pixel q to lane q%64.  Each lane adds its
 * elements in increasing e; the 64 lane sums are then combined by the balanced tree
 * v[i] += v[i^32], ^16, ^8, ^4, ^2, ^1.  (kroeger: Eigen .sum(), patch.cpp:74-76,178-179,278,331.) */
float dis_sum(const float *v, int n, int noc);

/* ---- patch grid (kroeger/patchgrid.cpp) ---- */
typedef struct dis_grid {
  int w, h, ps, noc, steps, nopw, noph, nop, pad, tmp_w, lvl;
  float lb, ubw, ubh;
  float *pt_ref;   /* nop x 2 */
  float *p_init;   /* nop x 2 */
  float *tmpl, *tdx, *tdy;   /* nop x novals */
  float *hes;      /* nop x 3: H00,H01,H11 */
  float *p_iter;   /* nop x 2 : result */
  float *pweight;  /* nop x novals */
  int   *cnt;      /* nop : iterations run */
  int depth;       /* copy of dis_params.depth */
  int camlr;       /* oflow.h:28, oflow.cpp:153,157: 0 for the forward grid (displacement <= 0), 1 for the backward grid (>= 0);
                      only read in depth mode (patch.cpp:188-193) */
} dis_grid;
dis_grid *dis_grid_new(int w, int h, int lvl, const dis_params *p);
void dis_grid_free(dis_grid *g);
void dis_grid_init(dis_grid *g, const dis_params *p, const float *I0, const float *I0x, const float *I0y);
void dis_grid_init_from_coarser(dis_grid *g, const float *flow_prev);
/* trace (optional): per patch (max_iter+1) x 4 floats [p0,p1,mares,cnt] rows; may be NULL */
void dis_grid_optimize(dis_grid *g, const dis_params *p, const float *I1, float *trace);
void dis_grid_aggregate(const dis_grid *g, const dis_params *p, float *flowout);
/* the same with a complementary grid cg (SetComplGrid, patchgrid.cpp:92-95): after g's own patches, every patch of cg is
 * splatted at its MOVED position with bilinear weights and reversed flow (patchgrid.cpp:278-375).  cg may be NULL. */
void dis_grid_aggregate_fb(const dis_grid *g, const dis_grid *cg, const dis_params *p, float *flowout);

/* ---- variational refinement (kroeger/refine_variational.cpp + FDF1.0.1) ----
 * planar images have stride = ceil4(w) like image_new (image.c:15-31). */
int dis_stride(int w);
void dis_image_warp(float *dst, float *mask, const float *src, const float *wx, const float *wy,
                    int w, int h, int noc);
void dis_get_derivatives(const float *im1, const float *im2w, int w, int h, int noc,
                         float *Ix, float *Iy, float *Iz, float *Ixx, float *Ixy, float *Iyy,
                         float *Ixz, float *Iyz);
void dis_compute_smoothness(float *horiz, float *vert, const float *uu, const float *vv,
                            int w, int h, float quarter_alpha);
void dis_compute_data(float *a11, float *a12, float *a22, float *b1, float *b2,
                      const float *mask, const float *du, const float *dv,
                      const float *Ix, const float *Iy, const float *Iz, const float *Ixx,
                      const float *Ixy, const float *Iyy, const float *Ixz, const float *Iyz,
                      int w, int h, int noc, float half_delta_over3, float half_gamma_over3);
void dis_sub_laplacian(float *dst, const float *src, const float *horiz, const float *vert,
                       int w, int h);
void dis_sor_coupled(float *du, float *dv, float *a11, float *a12, float *a22, const float *b1,
                     const float *b2, const float *horiz, const float *vert, int w, int h,
                     int iterations, float omega);
/* red-black ordering of the same 2x2 block update (throughput mode of the engine; NOT reference) */
/* FDF1.0.1/solver.c:19-72 sor_coupled_slow_but_readable, serial rows (a11 / a12 / a22 are read only) */
void dis_sor_coupled_slow(float *du, float *dv, const float *a11, const float *a12, const float *a22, const float *b1,
                          const float *b2, const float *horiz, const float *vert, int w, int h, int iterations, float omega);
void dis_sor_coupled_redblack(float *du, float *dv, float *a11, float *a12, float *a22,
                              const float *b1, const float *b2, const float *horiz,
                              const float *vert, int w, int h, int iterations, float omega);
/* whole VarRefClass ctor: I0/I1 padded level images, flow w x h x 2 interleaved, in place */
void dis_varref(const float *I0, const float *I1, int w, int h, int lvl, const dis_params *p,
                float *flow, int sor_mode /*0 lexicographic sor_coupled (reference default build), 1 red-black, 2 sor_coupled_slow_but_readable (reference OpenMP build, serial)*/);
/* stereo depth (SELECTMODE 2): compute_data_DE (opticalflow_aux.c:446-540), sor_coupled_slow_but_readable_DE
 * (solver.c:428-466, serial = lexicographic Gauss-Seidel) and RefLevelDE (refine_variational.cpp:243-330);
 * flow is w x h x 1, camlr selects the sign clamp of the update (:299-314) */
void dis_compute_data_de(float *a11, float *b1, const float *mask, const float *du,
                         const float *Ix, const float *Iy, const float *Iz, const float *Ixx,
                         const float *Ixy, const float *Iyy, const float *Ixz, const float *Iyz,
                         int w, int h, int noc, float half_delta_over3, float half_gamma_over3);
void dis_sor_de(float *du, const float *a11, const float *b1, const float *horiz, const float *vert,
                int w, int h, int iterations, float omega);
void dis_varref_depth(const float *I0, const float *I1, int w, int h, int lvl, const dis_params *p,
                      float *flow, int camlr);

/* ---- whole flow: OFClass ctor (kroeger/oflow.cpp:32-363) on prebuilt pyramids ---- */
/* flows (initflow, outflow, level_dump) have 2 channels, or 1 in depth mode */
void dis_flow_pyr(const dis_pyramid *P0, const dis_pyramid *P1, const dis_params *p,
                  const float *initflow, float *outflow, int sor_mode,
                  float *level_dump /* optional: concatenated pre/post-refinement flows */);
/* convenience: padded frames in -> finest-scale flow out (pyramid + flow) */
void dis_flow(const float *I0, const float *I1, int wp, int hp, const dis_params *p,
              float *outflow, int sor_mode);
/* bench.py's cpu_baseline, all-cores leg: n_total pairs (pair k = source pair k % nsrc of the unpadded w x h x noc frames)
 * frame-parallel on nthreads pthreads with thread-private block caches (no allocation per pair after a thread's first);
 * with_pyramid 1: padding + pyramids + flow per pair, 0: flow on pyramids built once per thread.  out (optional): the flows
 * of pairs 0 .. min(n_total, nsrc)-1.  Returns the seconds the n_total pairs took (every thread runs one untimed pair first). */
double dis_flow_many(const float *I0, const float *I1, long pair_stride, int nsrc, int w, int h, const dis_params *p,
                     int n_total, int nthreads, int with_pyramid, float *out);
/* parity-sensitivity switches (tests only): order of the per-patch reductions (definition D1) and of the 2x2 mean (D4) */
void dis_set_sum_order(int order);
void dis_set_mean_order(int order);
/* kroeger/run_dense.cpp:407-414: *2^sc_l, bilinear x2^sc_l (cv::resize INTER_LINEAR), crop */
void dis_upsample_crop(const float *flow, int wl, int hl, int sc_l, int padw, int padh,
                       int w_org, int h_org, float *out);
/* the same for nch interleaved channels (depth mode: 1, run_dense.cpp:387-388) */
void dis_upsample_crop_n(const float *flow, int wl, int hl, int sc_l, int padw, int padh,
                         int w_org, int h_org, int nch, float *out);

#ifdef __cplusplus
}
#endif
#endif
