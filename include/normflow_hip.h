/* normflow_hip.h -- C ABI of libnormflow_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the coupling-layer hot path of jkomijani/normflow_.
 * The reference has no FFI of its own (it is pure Python on eager aten ops), so
 * each entry point replaces an *eager op chain* of the reference; the file:line
 * of that chain is cited per function (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Rules of the boundary
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's
 *     data_ptr(), a hipMalloc'd buffer ...); the library never allocates, frees
 *     or synchronises; scratch space is passed in (`workspace`), its size is
 *     given by nf_workspace_bytes();
 *   - every launch goes to the `stream` argument (a hipStream_t passed as void*);
 *   - every function returns 0 on success, a negative NF_E* code otherwise, and
 *     never throws; nf_last_error_string() describes the last failure of the
 *     calling thread;
 *   - tensors are dense and row-major.  B = batch, V = number of lattice sites,
 *     C = channels of raw net output per site.  `dtype` selects the arithmetic
 *     and storage type of all floating tensors of a call.
 *
 * Site activity ("checkerboard masking", src/mask/mask.py:17-61) is given by a
 * byte mask of V entries: 1 = the site is transformed by this layer ("active"),
 * 0 = it is frozen.  Two layouts of the per-site parameter tensor exist:
 *   NF_LAYOUT_FULL : params is (B, C, V); entries at frozen sites are ignored;
 *   NF_LAYOUT_PAIR : params is (B, C, V/2): the sites are taken in aligned pairs
 *                    (2h, 2h+1), exactly one site of every pair is active (true
 *                    for an even-odd mask whose fastest axis is even), and column
 *                    h holds the parameters of that pair's active site.  This is
 *                    the HBM-efficient layout: no parameter bytes are read for
 *                    frozen sites.
 */
#ifndef NORMFLOW_HIP_H
#define NORMFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_VERSION 300 /* 0.3.0 */

/* NF_F16 (nf_rqs_fwd / nf_rqs_inv with knots_len 4/8/16, nf_affine_fwd / nf_affine_inv): x, params and y are IEEE half, the arithmetic is fp32 and
 * log0 / logj are fp32 ("fp16 params / fp32 log-det accumulate", BASELINE config 5). */
/* NF_F16_FIELD (nf_affine_fwd / nf_affine_inv only): x and y are IEEE half, params stay fp32 (what a conv layer wrote),
 * fp32 arithmetic and log-det. */
enum nf_dtype { NF_F32 = 0, NF_F64 = 1, NF_F16 = 2, NF_F16_FIELD = 3 };
enum nf_layout { NF_LAYOUT_FULL = 0, NF_LAYOUT_PAIR = 1 };
enum nf_extrap { NF_EXTRAP_NONE = 0, NF_EXTRAP_LINEAR = 1, NF_EXTRAP_ANTI = 2 };
enum nf_status {
  NF_OK = 0,
  NF_EINVAL = -1,    /* bad argument (null pointer, bad size, unsupported option) */
  NF_EWORKSPACE = -2,/* workspace too small */
  NF_ELAUNCH = -3    /* hip launch error */
};

/* Options of one rational-quadratic spline family
 * (RQSplineCoupling_.__init__, src/nn/scalar/couplings_.py:159-176). */
typedef struct nf_rqs_opts {
  int32_t m;            /* knots per spline; C = 3m-2, or 2m-1 / m with fixed knots */
  int32_t extrap_left;  /* nf_extrap */
  int32_t extrap_right; /* nf_extrap */
  int32_t layout;       /* nf_layout of `params` (and of grad_params) */
  double xlo, xhi, ylo, yhi;
  const void *fixed_knots_x; /* optional device array of m values (dtype), or NULL */
  const void *fixed_knots_y; /* optional device array of m values (dtype), or NULL */
} nf_rqs_opts;

/* Strides (in elements) from one batch entry to the next; 0 selects the dense
 * default.  They exist so that one data channel of a multi-channel tensor
 * (MultiRQSplineCoupling_, couplings_.py:279-436) can be addressed in place. */
typedef struct nf_strides {
  int64_t x_batch;      /* default V            */
  int64_t y_batch;      /* default V            */
  int64_t params_batch; /* default C * V or C*V/2 */
} nf_strides;

int nf_version(void);
const char *nf_last_error_string(void);

/* Process-wide kernel-selection options.  They choose between kernels that compute the SAME layer (results within the
 * tolerances stated per kernel); none of them skips work.  The library never reads the environment in a product build
 * (timing ablations and clock stamps exist only in `make DIAG=1` builds, which define NF_DIAG).
 *   NF_OPT_SPLIT16 (default 1): tanh / logistic ConvAct stacks may run the split-fp16 kernels (nf_conv_h.hip: every fp32
 *                  product as three fp16 matrix-core products, fp32 accumulation); 0 = exact fp32 MFMA products everywhere
 *                  (nf_conv_weight_layout / nf_conv_split16_supported answer accordingly);
 *   NF_OPT_PIPE    (default 1): eligible fp32 layers run the persistent, staging-overlapped kernels (nf_conv_pipe.hip);
 *                  0 = one box per workgroup (nf_conv.hip).
 *   NF_OPT_SMALL8  (default 1): the small-lattice fused layer (nf_small3d_rqs / nf_small_lattice_coupling) runs its
 *                  eight-wave form (two roles, two waves per SIMD); 0: the four-wave form (every wave holds all weights).
 * nf_set_option returns the previous value (>= 0) or NF_EINVAL; set options before launching, not concurrently with calls.
 */
enum nf_option { NF_OPT_SPLIT16 = 0, NF_OPT_PIPE = 1, NF_OPT_SMALL8 = 2, NF_OPT_COUNT_ = 3 };
int nf_set_option(int which, int value);
int nf_get_option(int which);

/* Bytes of scratch needed by any coupling / distconv call on a (B, V) problem. */
size_t nf_workspace_bytes(int64_t B, int64_t V);

/* ---- K2/K3: rational-quadratic spline coupling -----------------------------
 * Replaces, fused in one pass: knot construction (couplings_.py:211-262:
 * split, softmax, cumsum, scale/shift, softplus(beta=ln2)), boundary
 * augmentation (src/lib/spline/spline.py:458-532), bin search (:154-172),
 * segment evaluation (:185-220) or inversion (:222-287), purify + log +
 * per-sample sum (couplings_.py:186-188, src/nn/_core.py:38-42).
 *
 *   x       (B, V)  input field; only active sites are read
 *   params  (B, C, V) or (B, C, V/2) raw net output (logits)
 *   mask    (V) bytes, 1 = active
 *   log0    (B) or NULL (treated as 0)
 *   y       (B, V)  out: transformed value at active sites, 0 at frozen sites
 *   logj    (B)     out: log0 + sum over active sites of log|dy/dx|
 * nf_rqs_inv is the inverse map (y -> x) and adds log|dx/dy|; it uses the
 * cancellation-free root 2 a0 / (-a1 + sqrt(a1^2 - 4 a0 a2)).
 */
int nf_rqs_fwd(const void *x, const void *params, const uint8_t *mask, const void *log0,
               void *y, void *logj, int64_t B, int64_t V, const nf_rqs_opts *opts,
               const nf_strides *strides, void *workspace, size_t workspace_bytes,
               int dtype, void *stream);
int nf_rqs_inv(const void *y, const void *params, const uint8_t *mask, const void *log0,
               void *x, void *logj, int64_t B, int64_t V, const nf_rqs_opts *opts,
               const nf_strides *strides, void *workspace, size_t workspace_bytes,
               int dtype, void *stream);

/* Vector-Jacobian products of the two maps above (what autograd derives from the
 * eager chain in the reference; needed by Fitter.step, src/_normflowcore.py:288).
 *   grad_out  (B, V)  cotangent of the value output
 *   grad_logj (B)     cotangent of the log-Jacobian output
 *   grad_in   (B, V)  out: cotangent of the value input (0 at frozen sites)
 *   grad_params       out: same shape/layout as params (0 at frozen sites)
 * nf_rqs_fwd_vjp takes the forward INPUT x; nf_rqs_inv_vjp takes the inverse's
 * OUTPUT x (= the point on the x axis), so neither recomputes a root.
 */
int nf_rqs_fwd_vjp(const void *x, const void *params, const uint8_t *mask,
                   const void *grad_out, const void *grad_logj, void *grad_in,
                   void *grad_params, int64_t B, int64_t V, const nf_rqs_opts *opts,
                   const nf_strides *strides, int dtype, void *stream);
int nf_rqs_inv_vjp(const void *x, const void *params, const uint8_t *mask,
                   const void *grad_out, const void *grad_logj, void *grad_in,
                   void *grad_params, int64_t B, int64_t V, const nf_rqs_opts *opts,
                   const nf_strides *strides, int dtype, void *stream);

/* nf_affine_sites: nf_affine_fwd (inverse = 0) / nf_affine_inv (1) that additionally writes the log-derivative of every
 * site (-|s| forward, +|s| inverse; 0 for a shift layer and at frozen sites) to site_out (B,V) of dtype: what
 * Module_.sum_density passes through when propagate_density is set (src/nn/_core.py:19,38-42).  NF_F32 / NF_F64. */
int nf_affine_sites(const void *v, const void *params, const uint8_t *mask, const void *log0, void *out,
                    void *logj, void *site_out, int64_t B, int64_t V, int n_ch, int layout, int inverse,
                    void *workspace, size_t workspace_bytes, int dtype, void *stream);

/* ---- K2s: the spline object of a coupling layer -------------------------------
 * What a user of the reference reaches through RQSplineCoupling_.make_spline / _hack
 * (src/nn/scalar/couplings_.py:202-262) and through propagate_density
 * (src/nn/_core.py:19,38-42): the derivative of every site instead of only its summed log.
 *
 * nf_rqs_fwd_sites / nf_rqs_inv_sites are nf_rqs_fwd / nf_rqs_inv (same arguments, same y and
 * logj) and additionally write site_out (B,V) of dtype: with NF_SITES_LOG the log-derivative
 * log|dy/dx| (fwd) or log|dx/dy| (inv) of every active site, with NF_SITES_DERIVATIVE the
 * derivative itself (`g` of spline.py:87-123 with grad=True; the inverse returns 1/g as the
 * reference's spline.backward does, spline.py:222-287); 0 at frozen sites.  NF_F32 / NF_F64.
 *
 * nf_rqs_knots writes the knot tensors make_spline builds from the logits, before the boundary
 * augmentation of spline.py:458-532: knots (B, 3m, V) of dtype = knots_x (m planes), knots_y (m),
 * knots_d (m); fixed knots_x / knots_y are copied through.  params in the full layout (B, C, V).
 * The arithmetic is the coupling kernels' own: these are the knots they evaluate. */
enum nf_sites_mode { NF_SITES_LOG = 1, NF_SITES_DERIVATIVE = 2 };
int nf_rqs_fwd_sites(const void *x, const void *params, const uint8_t *mask, const void *log0,
                     void *y, void *logj, void *site_out, int site_mode, int64_t B, int64_t V,
                     const nf_rqs_opts *opts, const nf_strides *strides, void *workspace,
                     size_t workspace_bytes, int dtype, void *stream);
int nf_rqs_inv_sites(const void *y, const void *params, const uint8_t *mask, const void *log0,
                     void *x, void *logj, void *site_out, int site_mode, int64_t B, int64_t V,
                     const nf_rqs_opts *opts, const nf_strides *strides, void *workspace,
                     size_t workspace_bytes, int dtype, void *stream);
int nf_rqs_knots(const void *params, void *knots, int64_t B, int64_t V, const nf_rqs_opts *opts,
                 int dtype, void *stream);

/* ---- K2e: a spline from explicit knot tensors ------------------------------------
 * The reference's generic spline object RQSpline(knots_x, knots_y, knots_d) = Pade22Spline
 * (src/lib/spline/spline.py:39-68; forward / backward :87-123; searchsorted + clamp :154-172;
 * segment function :185-220; inverse :222-287, with the stable root of App. A #2).
 * v, out, deriv: (B, V) of dtype (deriv may be NULL).  A knot tensor is (B, K, V) planes, or a
 * shared vector of K entries when its shared_* flag is set (the reference's 1-D knots).  Knots
 * are evaluated as given (already augmented for extrapolation; values outside reuse the end
 * segments).  inverse: out = x(v), deriv = dx/dy = 1/g.  NF_F32 / NF_F64, B <= 65535. */
int nf_spline_eval(const void *v, const void *knots_x, const void *knots_y, const void *knots_d,
                   void *out, void *deriv, int64_t B, int64_t V, int K, int shared_x, int shared_y,
                   int shared_d, int inverse, int dtype, void *stream);

/* ---- K5h as a differentiable node: the fused last layer + RQ-spline coupling in a TRAINING step -----------------
 * Reference: Fitter.step (src/_normflowcore.py:275-294) differentiates couplings_.py:178-200 through autograd, which
 * keeps the (B, 3m-2, *L) logits of every layer alive.  Here the logits never exist in memory, forward or backward:
 *  nf_conv_rqs_split16_train = nf_conv_rqs on the split-fp16 kernel for hidden activations of unknown range:
 *    `in` = (B, 8, V) fp32 planes (fastest axis of 32 sites) or, with in_split16, the (B, V, 16) pair tensor
 *    nf_planes_to_split16 made of them; absmax_bits (nf_absmax_bits of the planes, or NULL = unit range): the input is /
 *    gets scaled by the matching power of two, the logits are descaled.  wsplit: NF_WLAYOUT_SPLIT16.  y (B, V), logj (B).
 *  nf_conv_rqs_split16_vjp: recomputes the logits in the kernel and sends the cotangents (grad_y (B, V), grad_logj (B))
 *    back through the spline: grad_logits (B, cout, V/2) fp32 pair-compact -- the form nf_conv_wgrad_split16 /
 *    nf_conv_dgrad_split16 read (compact_parity) -- and grad_x (B, V), zero at frozen sites.  x_point: the point on
 *    the x axis (the forward pass's input, or the inverse pass's output).  inverse: VJP of the inverse map.
 * knots_len 2..16, cout = 3m-2; lattices as nf_conv_rqs_split16_supported. */
int nf_conv_rqs_split16_train(const void *in, int in_split16, const void *wsplit, const void *bias, int cout,
                              const void *x_active, const void *log0, void *y, void *logj, int64_t B,
                              const int32_t *lattice, int active_parity, const void *absmax_bits,
                              const nf_rqs_opts *opts, int inverse, void *workspace, size_t workspace_bytes,
                              void *stream);
int nf_conv_rqs_split16_vjp(const void *in, int in_split16, const void *wsplit, const void *bias, int cout,
                            const void *x_point, const void *grad_y, const void *grad_logj, void *grad_logits,
                            void *grad_x, int64_t B, const int32_t *lattice, int active_parity,
                            const void *absmax_bits, const nf_rqs_opts *opts, int inverse, void *stream);

/* ---- K5s: a whole RQ-spline coupling layer of a SMALL 3-D lattice in one kernel ---------
 * Replaces, for lattices (L0, L1, 16) that fit a CU's LDS (16^3 = BASELINE config 3), the whole atom
 * src/nn/scalar/couplings_.py:178-200: net(x_frozen) = ConvAct 1 -> 8 -> 8 -> cout (src/nn/scalar/modules.py:120-145,
 * 3^3 circular kernels, tanh / logistic hidden activations), make_spline (:211-262), the spline map and log g
 * (src/lib/spline/spline.py:154-287), purify and sum_density -- the hidden activations and the logits never leave the CU.
 * x_frozen, x_active, y: (B, V) fp32 (x_frozen: zeros at the active sites; y: zeros at the frozen sites); log0 (B) or
 * NULL, logj (B).  w1 / w2 / w3: the layers' weights scaled by 2^10 and split into fp16 (hi, lo) MFMA fragments --
 *   w1 [hi|lo][64 lanes][8]: A operand, lane 16 g + co holds taps 8g .. 8g+7 (tap = 9 j0 + 3 j1 + j2, zero from 27 and
 *      for co >= 8) of the single input channel;
 *   w2 [kernel row 3 j0 + j1 (9)][hi|lo][64][8]: A operand, lane 16 g + 8 s + co holds the 8 input channels of tap g - s of
 *      the fastest axis for output site s of a pair (zero outside 0..2);
 *   w3 [column tile (3)][K slice (7)][hi|lo][64][8]: B operand, lane 16 g + n holds the 8 input channels of tap 4 slice + g
 *      for logit channel 16 tile + n (zero for tap 27 and channels >= cout);
 * b1, b2 (8), b3 (cout): fp32 biases or NULL.  Hidden widths below 8: zero-padded by the host.  cout = 3m - 2, m = 2..16.
 * active_parity: the active site of pair (2h, 2h+1) in row (z, y) is 2h + ((active_parity + z + y) & 1).  fp32 only. */
int nf_small3d_rqs_supported(const int32_t *lattice3, int cout, int m, int act1, int act2);
/* The same kernel for the other small-lattice atoms: kind 0 = RQ-spline coupling (cout = 3m - 2), kind 1 = AFFINE coupling
 * (couplings_.py:123-139: the net ends in (t, s), cout = 2; y = t + x e^{-|s|}, logj = log0 - sum |s|; opts may be NULL);
 * ndim 3: lattice (L0, L1, 16); ndim 2: lattice (L1, 16) -- BASELINE config 2's 16 x 16 -- with the 3^2 kernels embedded as the
 * middle plane (j0 = 1) of 3^3 weight tensors, zero elsewhere, packed as for nf_small3d_rqs.  In 2-D the active site of
 * pair (2h, 2h+1) in row y is 2h + ((active_parity + y) & 1). */
int nf_small_lattice_supported(const int32_t *lattice, int ndim, int kind, int cout, int m, int act1, int act2);
int nf_small_lattice_coupling(int kind, const void *x_frozen, const void *x_active, const void *w1, const void *b1,
                              const void *w2, const void *b2, const void *w3, const void *b3, const void *log0,
                              void *y, void *logj, int64_t B, const int32_t *lattice, int ndim, int active_parity,
                              int cout, int act1, int act2, const nf_rqs_opts *opts, int inverse, void *stream);
int nf_small3d_rqs(const void *x_frozen, const void *x_active, const void *w1, const void *b1, const void *w2,
                   const void *b2, const void *w3, const void *b3, const void *log0, void *y, void *logj,
                   int64_t B, const int32_t *lattice3, int active_parity, int cout, int act1, int act2,
                   const nf_rqs_opts *opts, int inverse, void *stream);

/* ---- K1: affine / shift coupling --------------------------------------------
 * Replaces couplings_.py:123-139 (affine: chunk, 2 purify, abs, exp, fma, sum)
 * and :110-116 (shift).  params is (B, 2, .) = (t, s) for affine, (B, 1, .) = t
 * for shift (n_ch selects).  fwd: y = t + x e^{-|s|}, logj = log0 - sum|s|;
 * inv: x = (y - t) e^{|s|}, logj = log0 + sum|s|.
 */
int nf_affine_fwd(const void *x, const void *params, const uint8_t *mask, const void *log0,
                  void *y, void *logj, int64_t B, int64_t V, int n_ch, int layout,
                  void *workspace, size_t workspace_bytes, int dtype, void *stream);
int nf_affine_inv(const void *y, const void *params, const uint8_t *mask, const void *log0,
                  void *x, void *logj, int64_t B, int64_t V, int n_ch, int layout,
                  void *workspace, size_t workspace_bytes, int dtype, void *stream);
/* `inverse` selects which map is differentiated; `v` is that map's input. */
int nf_affine_vjp(const void *v, const void *params, const uint8_t *mask,
                  const void *grad_out, const void *grad_logj, void *grad_in,
                  void *grad_params, int64_t B, int64_t V, int n_ch, int layout,
                  int inverse, int dtype, void *stream);

/* ---- K4: DistConvertor_ = Expit_ -> SplineNet_ (shared knots) -> Logit_ ------
 * Replaces src/nn/scalar/modules_.py:93-114, 277-302, 333-358 with ONE pass over
 * the field.  The spline is shared by all sites; its K knots, already augmented for
 * the boundary condition (src/lib/spline/spline.py:458-532; O(m) host-side work on
 * the learned logits), are staged in LDS by the kernels.
 *
 *   knots   3*K values of dtype: x[0..K) | y[0..K) | d[0..K), K <= 512
 *   stages  bit0 = expit before, bit1 = spline, bit2 = logit after
 *           (DistConvertor_ = 7, a bare SplineNet_ = 2)
 *   inverse runs the inverse chain (the reversed list of inverted stages,
 *           src/nn/_core.py:69-72)
 *   x, y    (B, V);  log0, logj (B)
 */
int nf_distconv(const void *x, const void *knots, int K, const void *log0, void *y, void *logj,
                int64_t B, int64_t V, int stages, int inverse, void *workspace,
                size_t workspace_bytes, int dtype, void *stream);
/* VJP: grad_in (B,V) and grad_knots (3K DOUBLES, summed over the whole field,
 * overwritten by the call).  `v` is the forward chain's input (inverse=0) or the
 * inverse chain's OUTPUT (inverse=1). */
int nf_distconv_vjp(const void *v, const void *knots, int K, const void *grad_out,
                    const void *grad_logj, void *grad_in, double *grad_knots, int64_t B,
                    int64_t V, int stages, int inverse, void *workspace,
                    size_t workspace_bytes, int dtype, void *stream);

/* ---- K5: circular 'same' convolution + bias + activation on the f32 matrix cores ----
 * Replaces one Conv{1,2,3}d(padding='same', padding_mode='circular') / Conv4d layer of
 * ConvAct together with the activation that follows it (src/nn/scalar/modules.py:120-145,
 * src/nn/scalar/convNd.py:86-126).  Lattices of dimension d < 4 are passed with leading
 * axes of extent 1 and kernel extent 1.
 *
 *   in      (B, cin, V) f32, channel planes (the layout of a torch (B, C, *L) tensor)
 *   wfrag   weights in MFMA-fragment order.  nf_conv_packed_steps(cin, ntaps) == 0 (cin a multiple
 *           of 4, or cin >= 8): [tap][ceil(cin/4)][ceil(cout/16)][4][16],
 *           wfrag[t][q][n][g][j] = W[16n + j][4q + g][t];
 *           otherwise (cin in {1,2,3,5,6,7}) K-packed: [step][ceil(cout/16)][4][16] with
 *           wfrag[s][n][g][j] = W[16n + j][kk % cin][kk / cin], kk = 4s + g, for
 *           s < nf_conv_packed_steps(cin, ntaps); 0 where out of range; taps in row-major
 *           kernel order
 *   bias    (cout) or NULL
 *   out     (B, cout, V), or with compact != 0 (B, cout, V/2): only the site of every
 *           aligned pair (2h, 2h+1) whose coordinate sum has parity `active_parity`
 *           (NF_LAYOUT_PAIR input of the coupling kernels)
 *   act     0 none, 1 tanh, 2 relu, 3 leaky_relu(0.01), 4 softplus, 5 abs, 6 sigmoid
 */
/* nf_conv_two_site(cout, compact, L3, k3) != 0: the layer is computed with two-site column packing
 * (cout <= 8): pass wfrag packed from the (16, cin, k0, k1, k2, k3+1) tensor W2 with
 * W2[o][..][t3] = W[o][..][t3] (t3 < k3) and W2[8+o][..][t3] = W[o][..][t3-1] (t3 >= 1), else 0. */
int nf_conv_two_site(int cout, int compact, int l3, int k3);
int nf_conv_cin_pad(int cin);
int nf_conv_ntiles(int cout);
int nf_conv_packed_steps(int cin, int ntaps);
/* `compact`: 0 = (B, cout, V) planes; 1 = pair-compact (B, cout, V/2), active sites only; NF_OUT_SPLIT16 (2) = the full
 * lattice as the fp16 (hi, lo) PAIR TENSOR of 8-output-channel two-site layers -- what nf_conv_fwd_split16 reads and writes
 * and nf_conv_rqs(flags = NF_CONV_UNIT_INPUT | NF_CONV_SPLIT16_INPUT) reads; same bytes as the 8 fp32 planes, (B, V, 16) halfs:
 *   per lattice row (the L3 sites of the fastest axis; rows in row-major order of the other axes) L3*32 bytes =
 *     [hi | lo][even sites | odd sites][L3/2 slots][8 channels], hi = fp16(a), lo = fp16(a - hi);
 *   the even block holds site 2s in slot s, the odd block site 2s - 1 (mod L3) in slot s.
 * A row is thus a contiguous image that LDS-DMA copies as it stands (L3 = 32: exactly one 1 KiB wave-instruction), every
 * 16-byte entry is one k-group of an MFMA A fragment, and a tap-by-tap read of the image is free of LDS bank conflicts. */
enum { NF_OUT_SPLIT16 = 2 };
int nf_conv_fwd(const void *in, const void *wfrag, const void *bias, void *out, int64_t B,
                const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act,
                int compact, int active_parity, int dtype, void *stream);
/* Weight layout nf_conv_fwd / nf_conv_rqs (fused != 0) expects for this layer; pure planning, no GPU work.
 *   NF_WLAYOUT_FRAGMENT (0): the fragment order above, [tap][cin_pad/4][ntiles][4][16];
 *   NF_WLAYOUT_ROWPACK  (1): [row][cin_pad/4][64][NV], row = the taps of all axes but the fastest (row-major),
 *       lane = 16*(channel within the quad) + (column within its tile), value index v = j3*ntiles + tile for
 *       tap j3 < K3 along the fastest axis (K3 = k3, or k3 + 1 with two-site packing), NV = K3*ntiles rounded
 *       up to a multiple of 4 (zero fill).  I.e. fragment order permuted so that everything a lane needs for one
 *       kernel row sits in NV/4 16-byte words.  Used by the persistent kernel (fp32, cin % 4 == 0, k3 == 3,
 *       <= 48 output columns).
 *   NF_WLAYOUT_SPLIT16  (2): fp16 pairs for the split-fp16 kernel, [column tile (3)][K slice (21)][hi|lo][64 lanes][8]:
 *       slice 7*j3 + i = tap j3 (fastest axis) of kernel rows 4i..4i+3 ((j0, j1, j2) row-major; row 27 = zeros);
 *       lane = 16*g + n holds, for column 16*tile + n, the 8 input channels of kernel row 4i + g;
 *       hi = fp16(1024 w), lo = fp16(1024 w - hi) (the factor 2^10 keeps the lo parts of typical weights out of
 *       fp16's subnormal range; the kernel scales the accumulators back).  Needs max|w| < 29.
 * `fused`: bit 0 = the layer is the fused last layer (nf_conv_rqs), bit 1 = NF_CONV_UNIT_INPUT will be passed, bit 2 =
 * NF_CONV_SPLIT16_INPUT will be passed (the split-fp16 kernel takes fastest axes other than 32 sites only from a pair tensor).
 * Returns the code, or -1 for invalid arguments. */
enum { NF_WLAYOUT_FRAGMENT = 0, NF_WLAYOUT_ROWPACK = 1, NF_WLAYOUT_SPLIT16 = 2 };
int nf_conv_weight_layout(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int compact,
                          int fused, int dtype);
/* A hidden 8 -> 8 layer whose input AND output are the fp16 (hi, lo) pair tensor above (nf_conv_g.hip: two-site columns,
 * one v_mfma_f32_16x16x32_f16 slice per kernel row, three fp16 products per fp32 product; persistent workgroups marching
 * 2 x 2 columns of lattice rows through an LDS ring filled by LDS-DMA).  in16 and out16 must not overlap.
 *   in16, out16: (B, V, 16) halfs; wsplit: [kernel row (27)][hi|lo][64 lanes][8] halfs -- lane 16*g + n holds, for column
 *   n = 8*shift + co, the 8 input channels of tap (g - shift) of that kernel row (zero outside 0..2), scaled by 2^10
 *   and split as in NF_WLAYOUT_SPLIT16; bias (8) fp32 or NULL; act must keep |out| <= 1 (tanh, logistic).
 *   nf_conv_split16_supported: 3^4 kernel, 8 -> 8 channels, a fastest axis of 32 + 16 n sites, even other extents. */
int nf_conv_split16_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int act);
/* The LAST layer of an AffineCoupling_'s net (8 -> 2 channels: t, s) fused with the coupling (src/nn/scalar/couplings_.py:123-139),
 * fed by a pair tensor: y = t + x e^{-|s|}, logj = log0 - sum|s| (inverse != 0: x = (y - t) e^{|s|}, + sum|s|) at the active
 * site of every pair, 0 at the frozen one; the (B, 2, V/2) parameter tensor never exists in memory.  Same kernel as
 * nf_conv_fwd_split16 with another epilogue.
 *   wsplit: the hidden-layer fragment format of nf_conv_fwd_split16 packed from the (8, 8, 3,3,3,3) tensor whose output
 *   channels 0, 1 are the layer's weights and 2..7 zero; bias (8) fp32 (entries 0, 1 used) or NULL; x_active, y (B, V) fp32 --
 *   or IEEE half with flags = NF_CONV_FIELD_F16; log0 (B) or NULL, logj (B): fp32; workspace as nf_workspace_bytes(B, V). */
int nf_conv_affine_split16(const void *in16, const void *wsplit, const void *bias, const void *x_active, const void *log0,
                           void *y, void *logj, int64_t B, const int32_t *lattice, int active_parity, int inverse,
                           int flags, void *workspace, size_t workspace_bytes, void *stream);
/* The FIRST ConvAct layer (1 -> 8 channels, 3^4 kernel, tanh / logistic) in front of the two entry points above (nf_conv_c.hip):
 * fp32 field in, fp16 pair tensor out, every fp32 product as three fp16 matrix-core products (x = x_hi + x_lo, |x| below the
 * fp16 range: beyond 6.5e4 the outputs are NaN, never silently wrong).
 *   in (B, V) fp32 (the frozen half of the field, one channel); out16 (B, V, 16) halfs; bias (8) fp32 or NULL;
 *   wsplit: [K slice (4)][hi|lo][64 lanes][8] halfs: K index = 4 r + t, r = kernel row (j0, j1, j2) row-major (27, padded to
 *   32 with zeros), t = tap of the site pair (sites 2p-1 .. 2p+2); lane 16*g + n (column n = 8*shift + co) holds rows
 *   8*slice + 2*g + h (h = 0, 1) as values 4*h + t = W[co][r][t - shift] (zero outside 0..2), scaled by 2^10 and split as
 *   in NF_WLAYOUT_SPLIT16.  nf_conv_first_split16_supported: 8 output channels, 3^4 kernel, a fastest axis of 32 + 16 n
 *   sites, even other extents. */
int nf_conv_first_split16_supported(const int32_t *lattice, const int32_t *ksize, int cout, int act);
int nf_conv_first_split16(const void *in, const void *wsplit, const void *bias, void *out16, int64_t B,
                          const int32_t *lattice, int act, void *stream);
int nf_conv_fwd_split16(const void *in16, const void *wsplit, const void *bias, void *out16, int64_t B,
                        const int32_t *lattice, int act, void *stream);
/* Which kernel the calling thread's last nf_conv_fwd / nf_conv_rqs launched: 0 = one box per workgroup
 * (nf_conv.hip), 1 = persistent workgroups with staging overlapped with the MFMAs (nf_conv_pipe.hip;
 * fp32, cin % 4 == 0, kernel extent 3 on the fastest axis), 2 = the single-input-channel kernel of the first
 * ConvAct layer (nf_conv_pipe.hip, conv_c1_kernel).  Same results either way; for tests and benches. */
int nf_conv_last_path(void);

/* ---- K5+K2 fused: last conv layer of the parameter net + RQ-spline coupling ---------------
 * The (B, 3m-2, V/2) logit tensor is produced in the MFMA accumulators, staged in LDS and
 * consumed by the spline map in the same workgroup; it never goes to HBM (SURVEY 8(f) item 1:
 * src/nn/scalar/modules.py:120-145 feeding src/nn/scalar/couplings_.py:178-200).
 * Inference only (no VJP); needs a plain even-odd activity pattern (`active_parity` = parity of
 * the coordinate sum of the active sites), an even fastest axis, knots_len in {4, 8, 16}.
 *   in (B, cin, V) hidden activations; wfrag/bias as nf_conv_fwd with cout = 3m-2;
 *   x_active, y (B, V); log0, logj (B); inverse != 0 applies the inverse map.
 *   flags: NF_CONV_UNIT_INPUT = the caller guarantees |in| <= 1 (hidden activations that are tanh / sigmoid
 *   outputs): eligible layers (8 -> 46 channels, 3^4 kernel, a fastest axis of 32 sites -- or, fed by a pair tensor, 32 + 16 n --, even
 *   other extents) then run the split-fp16 kernel
 *   (nf_conv_h.hip): every fp32 product as three fp16 matrix-core products with fp32 accumulation, ~1.7x the
 *   rounding error of an fp32 chain, well inside the 1e-5 budget; the weights must then be packed in
 *   NF_WLAYOUT_SPLIT16 (nf_conv_weight_layout with `fused` = 1 | 2 says which layout a layer wants).
 */
/*   NF_CONV_SPLIT16_INPUT (with NF_CONV_UNIT_INPUT): `in` is not (B, 8, V) fp32 but what the previous layer wrote
 *   with nf_conv_fwd(compact = NF_OUT_SPLIT16): (B, V, 16) IEEE halfs = per site hi[8] | lo[8]. */
/*   NF_CONV_FIELD_F16: x_active and y are IEEE half (fp16 field storage; the arithmetic stays fp32 and log0 / logj stay fp32:
 *   BASELINE config 5, "fp16 params / fp32 log-det accumulate" -- in the fused path the "params" never exist in memory). */
enum { NF_CONV_UNIT_INPUT = 1, NF_CONV_SPLIT16_INPUT = 2, NF_CONV_FIELD_F16 = 4 };
int nf_conv_rqs_supported(int cout, int m);
/* Planning query for the split-fp16 chain: its fused last layer takes ANY knots_len 2 <= m <= 16 (cout = 3m - 2 <= 46
 * logit channels: the reference leaves knots_len free, src/nn/scalar/couplings_.py:211-262) on 4-D lattices with even
 * extents and a fastest axis of 32 + 16 n sites, fed by the pair tensor (NF_CONV_UNIT_INPUT | NF_CONV_SPLIT16_INPUT).
 * Hidden widths below 8 run the same kernels on weights zero-padded to 8 channels (the host packs them). */
int nf_conv_rqs_split16_supported(const int32_t *lattice, int cout, int m);
int nf_conv_rqs(const void *in, const void *wfrag, const void *bias, const void *x_active,
                const void *log0, void *y, void *logj, int64_t B, const int32_t *lattice,
                const int32_t *ksize, int cin, int cout, int active_parity,
                const nf_rqs_opts *opts, int inverse, int flags, void *workspace, size_t workspace_bytes,
                int dtype, void *stream);

/* ---- end points of the flow ------------------------------------------------
 * nf_phi4_action: S[b] = sum_x (w2 phi^2 + w4 phi^4) - w0 sum_mu sum_x phi(x) phi(x - mu), periodic
 * (ScalarPhi4Action.action, src/action/scalar_action.py:38-46; w0, w2, w4 from get_coef, :24-36):
 * ONE pass instead of the reference's d+1.  cfgs (B, V), lattice[4] (leading extents 1 for d < 4),
 * action (B).  nf_phi4_action_vjp: grad_cfgs = grad_action[b] * dS/dphi.
 * nf_normal_logprob: logp[b] = sum_x [-(x-loc)^2/(2 s^2) - log s - log sqrt(2 pi)], loc/scale (V)
 * or NULL (0 / 1) (Prior.log_prob with torch.distributions.Normal, src/prior/prior.py:30-36).
 */
int nf_phi4_action(const void *cfgs, void *action, int64_t B, const int32_t *lattice, double w0, double w2,
                   double w4, void *workspace, size_t workspace_bytes, int dtype, void *stream);
int nf_phi4_action_vjp(const void *cfgs, const void *grad_action, void *grad_cfgs, int64_t B,
                       const int32_t *lattice, double w0, double w2, double w4, int dtype, void *stream);
int nf_normal_logprob(const void *x, const void *loc, const void *scale, void *logp, int64_t B, int64_t V,
                      void *workspace, size_t workspace_bytes, int dtype, void *stream);
int nf_normal_logprob_vjp(const void *x, const void *loc, const void *scale, const void *grad_logp,
                          void *grad_x, int64_t B, int64_t V, int dtype, void *stream);
/* Folded (XOR) into the high key word of nf_normal_sample's Philox generator: separates its streams from torch's own
 * Philox kernels, which key on the bare seed. */
#define NF_PHILOX_KEY_DOMAIN 0x6e66686bu   /* 'nfhk' */

/* nf_normal_sample: Prior.sample_ for a NormalPrior (src/prior/prior.py:26-29 with :30-36 and :92-101) in ONE launch:
 * x[b, i] = loc[i] + scale[i] z[b, i] with z standard normal, and logr[b] = sum_i [-z^2/2 - log scale[i] - log sqrt(2 pi)]
 * accumulated from the z still in registers (the reference draws, then re-reads the field for log_prob, then sums).
 * Generator: Philox4x32-10 (Salmon et al., SC'11; the Random123 / cuRAND / torch generator), counter-based and stateless:
 *   group q = i / 4 (float32: four normals per call) or i / 2 (float64: two), g = b * ceil(V / per) + q,
 *   counter = (lo32 g, hi32 g, lo32 offset, hi32 offset), key = (lo32 seed, hi32 seed), outputs r0..r3;
 *   float32: (z0, z1) = rho (cos, sin)(2 pi u2) with u1 = (r0 + 1) 2^-32, u2 = r1 2^-32, rho = sqrt(-2 ln u1); (z2, z3) from (r2, r3);
 *   float64: u1 = ((r0 << 21 ^ r1 >> 11) + 1) 2^-53, u2 = (r2 << 21 ^ r3 >> 11) 2^-53.
 * The caller owns the stream position: advance `offset` by 1 per call (the host mirror takes seed and offset from torch's
 * CUDA generator, so torch.manual_seed governs this kernel too).  loc / scale: (V) or NULL (0 / 1). */
int nf_normal_sample(void *x, void *logr, const void *loc, const void *scale, int64_t B, int64_t V, uint64_t seed,
                     uint64_t offset, void *workspace, size_t workspace_bytes, int dtype, void *stream);

/* ---- VJP of the conv layer (K5) ---------------------------------------------------------------
 * grad_input is nf_conv_fwd itself applied to the pre-activation cotangent with the weights
 * flipped along every kernel axis and in/out channels swapped.  The two entry points below are
 * the rest of what autograd derives for ConvAct in Fitter.step (src/_normflowcore.py:288):
 *   nf_act_vjp     grad_pre = grad_out * act'(.), the derivative written through the activation's
 *                  OUTPUT y (tanh: 1-y^2, relu/leaky: sign test, softplus: 1-e^-y, sigmoid: y(1-y));
 *   nf_conv_wgrad  gw[o][t*cin + i] += sum_{b,n} gz[b,o,n] * in[b,i,(n + t - k//2) mod L] and
 *                  gw[o][ntaps*cin] += sum_{b,n} gz[b,o,n] (the bias gradient), accumulated with
 *                  float atomics into a ZEROED (16*ceil(cout/16), nf_conv_wgrad_cols(cin, ntaps))
 *                  buffer; cout <= 48 per call; gz is the full-lattice (B, cout, V) cotangent.
 */
int nf_act_vjp(const void *grad_out, const void *y, void *grad_pre, int64_t n, int act, int dtype,
               void *stream);
int nf_conv_wgrad_cols(int cin, int ntaps);
int nf_conv_wgrad(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice,
                  const int32_t *ksize, int cin, int cout, int dtype, void *stream);

/* The same weight gradient on the fp16 matrix cores (every fp32 product as three fp16 products, fp32 accumulation;
 * gz scaled by the power of two of *absmax_bits, see nf_absmax_bits) for the lattice networks' shapes: 4-D lattice with 32 + 16 n sites on
 * the fastest axis, 3^4 kernels, cin 1 or 8, cout <= 48, fp32.  Same gw layout and accumulate-into-gw semantics as
 * nf_conv_wgrad; deterministic (per-workgroup partial matrices in the workspace, summed in a fixed order).
 * compact_parity < 0: gz is the full-lattice (B, cout, V) tensor; 0 / 1: gz is pair-compact (B, cout, V/2), the cotangent of
 * an active-site-only layer (coordinate sum == compact_parity mod 2), read as it is -- no expanded copy (same for
 * nf_planes_to_split16).
 * nf_conv_wgrad_split16_supported says whether a shape qualifies; callers fall back to nf_conv_wgrad otherwise. */
/* Input gradients of the lattice networks' conv layers on the split-fp16 chain (training).  The gradient w.r.t. the input of
 * a layer with 8 input channels is a convolution of the output cotangent gz (B, C, V) with the flipped, transposed weights:
 * nf_planes_to_split16 writes gz, scaled into fp16's range by the power of two that belongs to *absmax_bits (nf_absmax_bits;
 * NULL: unscaled, for activations), as ceil(C / 8) pair tensors (G, B, V, 16 halfs; layout of NF_OUT_SPLIT16);
 * nf_conv_dgrad_split16 runs the hidden-layer kernel on one of them with weights packed like a forward 8 -> 8 layer's
 * (flipped / transposed by the caller, zero rows past C), bias NULL, act 0, and writes -- or, with accumulate, adds --
 * the descaled fp32 planes gx (B, 8, V).  The same entry runs a FORWARD 8 -> 8 layer whose output training keeps as planes:
 * activations as an unscaled pair tensor (absmax_bits NULL), bias, act = tanh.  Lattices as nf_conv_split16_supported. */
int nf_planes_to_split16(const void *gz, void *out16, const void *absmax_bits, int64_t B, int C,
                         const int32_t *lattice, int compact_parity, void *stream);
int nf_conv_dgrad_split16(const void *in16, const void *wsplit, const void *bias, void *gx, int64_t B,
                          const int32_t *lattice, const void *absmax_bits, int accumulate, int act, void *stream);

/* The last conv layer 8 -> 46 of a spline coupling's net at the active sites on the split-fp16 kernel (the matrix-core part of
 * nf_conv_rqs), with the logits written out pair-compact (B, 46, V/2) fp32 instead of consumed: the forward pass of a
 * training step, which differentiates the spline separately.  in: fp32 channel planes (B, 8, V) (fastest axis of 32 sites)
 * or, with in_split16, the (B, V, 16) pair tensor; absmax_bits (nf_absmax_bits of the input, or NULL for inputs already in
 * fp16's range): the input gets / is scaled by the matching power of two and the logits are descaled; wsplit in
 * NF_WLAYOUT_SPLIT16; bias (46) fp32 or NULL. */
int nf_conv_last_logits_split16(const void *in, int in_split16, const void *wsplit, const void *bias, void *logits,
                                int64_t B, const int32_t *lattice, int active_parity, const void *absmax_bits,
                                void *stream);
/* ... the same for any cout <= 46 (logits (B, cout, V/2); weights still packed to 48 columns), ADDED to the logits already in
 * `logits` when accumulate != 0 (bias NULL then): the second group of 8 input channels of a 16 -> cout layer -- hidden width 16 on the split-fp16 kernels (reference: modules.py:68-154 leaves the hidden
 * widths free); normflow__amd/_hip.py, conv_wide_logits_split16 composes the stack 1 -> 16 -> 16 -> 3m-2 from
 * nf_conv_first_split16 x 2, nf_conv_dgrad_split16 x 4 (+ nf_planes_to_split16 x 2) and this entry x 2. */
int nf_conv_last_logits_split16_acc(const void *in, int in_split16, const void *wsplit, const void *bias, void *logits,
                                    int64_t B, const int32_t *lattice, int active_parity, const void *absmax_bits,
                                    int accumulate, int cout, void *stream);

/* nf_expand_pairs: a pair-compact tensor (rows, V/2) -- the layout the active-site-only conv output and its cotangent use --
 * to the full lattice (rows, V), zeros at the sites of the other parity; rows = B * channels; lattice[3] even. */
int nf_expand_pairs(const void *compact, void *full, int64_t rows, const int32_t *lattice, int parity, int dtype,
                    void *stream);
/* nf_gather_pad: out[i] = index[i] in [0, nsrc) ? src[index[i]] : 0 for i < n, elements of 2, 4 or 8 bytes moved as bits.
 * The host side re-arranges a layer's weights into a kernel's fragment layout with it (normflow__amd/_hip.py, _pack_by_gather:
 * the index map of a layer shape is built once): one launch where the torch expression of the same packing is a dozen. */
int nf_gather_pad(const void *src, const int32_t *index, void *out, int64_t n, int64_t nsrc, int elem_bytes, void *stream);
/* nf_conv_wgrad_sites: the same gradient (same gw layout and accumulate-into-gw contract as nf_conv_wgrad) for layers with
 * FEW columns -- taps x cin + 1 <= 224, i.e. every 1-, 2- and 3-D 3-tap layer of up to 8 input channels: the layers of the
 * small lattices flows are usually trained on (reference: src/_normflowcore.py:275-294 differentiating
 * src/nn/scalar/modules.py:120-145).  Every wave holds all column tiles and the waves split the sites of a box; no atomics --
 * per-workgroup partial matrices in `workspace` (nf_conv_wgrad_sites_workspace bytes) are added to gw in a fixed order, so the
 * gradient is bitwise reproducible.  compact_parity 0 / 1: gz is the pair-compact (B, cout, V/2) cotangent of an active-site-only
 * layer (values at the sites whose coordinate sum == parity mod 2; needs an even fastest axis) and only those sites are
 * walked; -1: gz is (B, cout, V).  fp32 and fp64 (fp64 with cout > 32 only up to 96 columns). */
int nf_conv_wgrad_sites_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout, int dtype);
size_t nf_conv_wgrad_sites_workspace(const int32_t *ksize, int cin, int cout, int dtype);
int nf_conv_wgrad_sites(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice, const int32_t *ksize,
                        int cin, int cout, int compact_parity, void *workspace, size_t workspace_bytes, int dtype,
                        void *stream);
int nf_conv_wgrad_split16_supported(const int32_t *lattice, const int32_t *ksize, int cin, int cout);
size_t nf_conv_wgrad_split16_workspace(int64_t B, const int32_t *lattice, int cin);
int nf_conv_wgrad_split16(const void *in, const void *gz, void *gw, int64_t B, const int32_t *lattice,
                          const int32_t *ksize, int cin, int cout, const void *absmax_bits, int compact_parity,
                          void *workspace, size_t workspace_bytes, void *stream);
/* max |x| of an fp32 tensor of n elements, as the bits of a float in 4 bytes of device memory: what the training kernels
 * (nf_conv_wgrad_split16, nf_planes_to_split16 / nf_conv_dgrad_split16) scale a cotangent by; one pass serves both. */
int nf_absmax_bits(const void *x, int64_t n, void *bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NORMFLOW_HIP_H */
