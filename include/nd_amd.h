/*
 * nd_amd.h -- C ABI of libnd_amd.so, the MI355X (gfx950) implementation of the
 * per-pixel compute path of jnhansen/nd.
 *
 * Every entry point replaces one native (or third-party native) call that the
 * reference's Python layer makes; the reference-side binding a maintainer
 * would add is shown in INTEGRATION.md.  All pointers named `*_dev` / data
 * pointers are DEVICE pointers (HIP), all sizes/strides are in ELEMENTS unless
 * they say bytes.  Calls enqueue work on `hip_stream` and return without
 * synchronising (exceptions are noted per function).  Return value: 0 on
 * success, a negative ND_AMD_E* code otherwise; nd_amd_last_error() returns a
 * thread-local description.  The library never throws and never aborts.
 *
 * No host twins.  SURVEY.md 8(b) sketched a `*_cpu` entry next to every device
 * entry (same arguments on host pointers).  The ABI deliberately has none: the
 * reference interface itself has no such pair (its native calls ARE the host
 * path), the product path has no CPU fallback by rule -- every entry fails with
 * ND_AMD_EINVAL / ND_AMD_EHIP rather than compute on the host -- and the only
 * CPU arithmetic of this repository, oracle/, is test infrastructure that
 * nothing under nd_amd/ may link or call.  A caller that wants the host path
 * keeps calling the reference's own nd._change / nd._filters / scipy.
 *
 * Paths are relative to the reference checkout (/root/reference).
 */
#ifndef ND_AMD_H
#define ND_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ND_AMD_ABI_VERSION 1

/* dtype of the `floating` fused type (nd/_change.pyx:8, nd/_filters.pyx:3) */
#define ND_AMD_F32 0
#define ND_AMD_F64 1

#define ND_AMD_OK            0
#define ND_AMD_EINVAL       -1   /* bad argument */
#define ND_AMD_EWORKSPACE   -2   /* workspace missing or too small */
#define ND_AMD_EHIP         -3   /* a HIP runtime call failed */
#define ND_AMD_ENOSOLUTION  -4   /* nlmeans: find_weight has no solution (ValueError in the reference) */
#define ND_AMD_EUNSUPPORTED -5   /* valid request this build does not cover */

/* scipy.ndimage border modes accepted by nd_amd_correlate (nd/filters.py:226
 * forwards **kwargs to scipy.ndimage.convolve; default 'reflect') */
#define ND_AMD_MODE_REFLECT  0
#define ND_AMD_MODE_CONSTANT 1
#define ND_AMD_MODE_NEAREST  2
#define ND_AMD_MODE_MIRROR   3
#define ND_AMD_MODE_WRAP     4

/* ids reported by nd_amd_timing_collect */
#define ND_AMD_KERNEL_OMNIBUS_GLOBAL 1
#define ND_AMD_KERNEL_OMNIBUS_SEARCH 2
#define ND_AMD_KERNEL_CORRELATE      3
#define ND_AMD_KERNEL_NLMEANS        4
#define ND_AMD_KERNEL_BOXCAR_TILED   5
#define ND_AMD_KERNEL_NLMEANS_TILED  6
#define ND_AMD_KERNEL_CORRELATE1D    7
#define ND_AMD_KERNEL_RELAYOUT       8
#define ND_AMD_KERNEL_OMNIBUS_DENSE  9
#define ND_AMD_KERNEL_OMNIBUS_FUSED  10   /* pass A with the change-point search fused in */
#define ND_AMD_KERNEL_OMNIBUS_SAMPLE 11   /* density sample that gates the fused form */
#define ND_AMD_KERNEL_OMNIBUS_EXACT  12   /* pass B, exact form behind the register form (marked pixels only) */

int nd_amd_abi_version(void);
const char *nd_amd_last_error(void);

/* ------------------------------------------------------------------------
 * OmnibusTest, dual-pol C2.
 * Replaces  nd._change.change_detection(values, alpha, n, njobs)
 *           nd/_change.pyx:263-287, sole caller nd/change.py:69.
 *
 * The reference passes one (y, x, time, 4) strided view whose last axis is
 * [C11, C12__re, C12__im, C22] (nd/change.py:66-67); here the four variables
 * are four plane pointers that share one set of element strides, which is
 * what that view is in memory.  Fast path: stride_x == 1 and 16-byte aligned
 * rows (planar [time][y][x]); any other strides run the generic kernel.
 *
 *   change   : uint8 (y, x, time) C-order, caller-allocated, fully overwritten
 *              (the reference returns a fresh np.zeros array, :275).
 *   z_out,
 *   p_out    : optional (y, x) rasters of dtype `dtype`: the test statistic
 *              -2 rho ln Q and the probability P of the global test over the
 *              whole series (nd/_change.pyx:46-77, 133-151) -- values the
 *              reference only exposes per pixel via its cpdef functions.
 *   workspace: device scratch, 256-byte aligned.  It holds the list of pixels
 *              whose global test can fire and a compact copy of their series
 *              (so the change-point search never re-reads the planes).
 *              nd_amd_omnibus_c2_workspace_bytes() returns the recommended
 *              size (room for the series of 1/8 of the pixels) and, through
 *              *min_bytes, the smallest size the call accepts; anything in
 *              between trades speed on change-rich rasters for memory.
 * njobs has no equivalent: the whole raster is one launch.
 * Series length: any k >= 1; parity-tested up to k = 193.  The first call
 * with a new (k, n_looks, alpha, dtype) tabulates one pair of decision bounds
 * per sub-series length on the host (O(k^2) work, cached); their safety
 * margin grows with k (see omni_bounds) so that very long series stay exact.
 * Fast forms of the regimes in which most pixels change (alpha below ~0.9;
 * chosen by threshold, series length and a device-side sample of the data)
 * cover k <= 192 (as do those of the sparse regime); longer series are still
 * exact but search pixel by pixel.  Where no screen can decide the test over
 * the whole series (omega2 outside [0, 1]: the reference's default n_looks = 1
 * on more than a few dates) that one test is evaluated exactly for every pixel
 * in the first pass, and only the pixels it fires for are searched.
 * ---------------------------------------------------------------------- */
size_t nd_amd_omnibus_c2_workspace_bytes(int dtype, int64_t ny, int64_t nx, int64_t k,
                                         size_t *min_bytes);

int nd_amd_omnibus_c2(const void *c11, const void *c12re, const void *c12im,
                      const void *c22, int dtype,
                      int64_t ny, int64_t nx, int64_t k,
                      int64_t stride_y, int64_t stride_x, int64_t stride_t,
                      uint32_t n_looks, double alpha,
                      uint8_t *change, void *z_out, void *p_out,
                      void *workspace, size_t workspace_bytes,
                      void *hip_stream);

/* ------------------------------------------------------------------------
 * OmnibusTest(ml=w): spatial multilooking fused into the test.
 * Replaces  ds_m = BoxcarFilter(w=ml).apply(ds_m); n = ml ** 2
 *           followed by nd._change.change_detection(values, alpha, n)
 *           nd/change.py:61-69 (BoxcarFilter = scipy.ndimage.convolve with
 *           ones((ml, ml)) / ml**2, mode 'reflect': nd/filters.py:256-267, 294-298).
 *
 * The four planes are read once: every (date, variable) value is the boxcar
 * mean of its ml x ml window in scipy's arithmetic (double products and sums
 * in footprint order, rounded to float32) and exists only in registers; the
 * number of looks is ml * ml.  Results equal nd_amd_correlate on every plane
 * followed by nd_amd_omnibus_c2 bit for bit.
 * Covered: float32, stride_x == 1, ml = 3 or 5, 2 <= k <= 24, ny, nx >= ml;
 * anything else returns ND_AMD_EUNSUPPORTED (and the workspace query 0): the
 * caller then multilooks with nd_amd_correlate and calls nd_amd_omnibus_c2.
 * The workspace must hold the multilooked series of every pixel the test can
 * list (they exist nowhere else): nd_amd_omnibus_c2_ml_workspace_bytes() is
 * the minimum the call accepts; only the listed part is ever touched.
 * ---------------------------------------------------------------------- */
size_t nd_amd_omnibus_c2_ml_workspace_bytes(int dtype, int64_t ny, int64_t nx, int64_t k, int ml);

int nd_amd_omnibus_c2_ml(const void *c11, const void *c12re, const void *c12im,
                         const void *c22, int dtype,
                         int64_t ny, int64_t nx, int64_t k,
                         int64_t stride_y, int64_t stride_x, int64_t stride_t,
                         int ml, double alpha,
                         uint8_t *change, void *z_out, void *p_out,
                         void *workspace, size_t workspace_bytes,
                         void *hip_stream);

/* ------------------------------------------------------------------------
 * OmnibusTest, full-pol C3 (3 x 3 complex Hermitian) -- EXTENSION.
 * The reference has no full-pol implementation (p = 2 is hard-coded,
 * nd/_change.pyx:51, 99, 135); this is the same algorithm with p = 3 and the
 * generic `_f`, `_rho`, `_omega2` of nd/_change.pyx:20-39 (BASELINE.json
 * config "OmnibusTest full-pol C3").  planes[9], all with the same element
 * strides: C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im.
 * Everything else as nd_amd_omnibus_c2; k <= 96.
 * Workspace: candidate lists plus, for series of up to 64 dates, room for
 * the series of one pixel in 16 (float32: the sparse regime's pass A hands
 * the candidates' series to the search from its registers; what does not
 * fit is gathered from the planes, slower, never wrong): ~ 6.5 % of the
 * size of a float32 stack.
 * ---------------------------------------------------------------------- */
size_t nd_amd_omnibus_c3_workspace_bytes(int64_t ny, int64_t nx, int64_t k);

int nd_amd_omnibus_c3(const void *const planes[9], int dtype,
                      int64_t ny, int64_t nx, int64_t k,
                      int64_t stride_y, int64_t stride_x, int64_t stride_t,
                      uint32_t n_looks, double alpha,
                      uint8_t *change, void *z_out, void *p_out,
                      void *workspace, size_t workspace_bytes,
                      void *hip_stream);

/* ------------------------------------------------------------------------
 * nd_amd_omnibus_c3 for data in the reference's own layout, without a
 * transpose (the full-pol counterpart of nd_amd_omnibus_c2_pixel_major;
 * nd/change.py:66-67 stacks (y, x, time) variables): plane c holds element
 * (y, x, t) at  planes[c][((y * nx + x) * k + t) * date_stride[c]].
 * date_stride = 1 for a real (y, x, time) array; 2, with
 * planes[c + 1] == planes[c] + 1, for the two halves of an interleaved
 * complex C12 / C13 / C23.  Accepted: nine real arrays, or three real and
 * three interleaved complex ones; 16-byte aligned; k a multiple of 4
 * (float64: of 2) with 9 k elements of 16 pixels within 56 KB; the sparse
 * regime, alpha >= 0.75.  Everything else returns ND_AMD_EUNSUPPORTED --
 * transpose and call nd_amd_omnibus_c3.  The series is folded out of LDS
 * images of the contiguous per-pixel runs, and the search reads a listed
 * pixel's series where it lies (9 runs of k values instead of 9 k values
 * gathered from planes).  Workspace as for nd_amd_omnibus_c3.
 * ---------------------------------------------------------------------- */
int nd_amd_omnibus_c3_pixel_major(const void *const planes[9], int dtype,
                                  int64_t ny, int64_t nx, int64_t k,
                                  const int64_t date_stride[9],
                                  uint32_t n_looks, double alpha, uint8_t *change,
                                  void *z_out, void *p_out,
                                  void *workspace, size_t workspace_bytes, void *hip_stream);

/* ------------------------------------------------------------------------
 * Kernel convolution / boxcar.
 * Replaces  scipy.ndimage.convolve(arr, nd_kernel, output=output, **kwargs)
 *           as called at nd/filters.py:256-267 (scipy is the reference's
 *           third-party arithmetic for ConvolutionFilter / BoxcarFilter).
 *
 * The host (nd_amd/filters.py) turns the kernel into scipy's footprint: the
 * non-zero taps (|w| > DBL_EPSILON) of the flipped kernel in C order, with
 * per-axis input offsets.  The array is viewed as 4-D (dims[4], missing
 * leading axes = 1) with element strides.  Per output element:
 *   double tmp = 0; for each tap: tmp += w * (double)in[extend(i + off)];
 *   out = (T)tmp          -- same order, double accumulation, no FMA.
 *   offsets : host pointer, ntaps x 4 int64
 *   weights : host pointer, ntaps double
 * Up to 128 taps travel to the kernel as a launch argument (fully asynchronous).
 * Larger footprints are copied into `taps_dev`
 * (>= 24 * ntaps bytes of device memory, may be NULL otherwise); that path
 * synchronises the stream once.
 * ---------------------------------------------------------------------- */
int nd_amd_correlate(const void *in, void *out, int dtype,
                     const int64_t dims[4],
                     const int64_t in_strides[4], const int64_t out_strides[4],
                     int64_t ntaps, const int64_t *offsets, const double *weights,
                     int mode, double cval,
                     void *taps_dev, size_t taps_dev_bytes,
                     void *hip_stream);

/* ------------------------------------------------------------------------
 * 1-D correlation along one axis -- the building block of GaussianFilter.
 * Replaces  scipy.ndimage.correlate1d(input, weights, axis, output, mode, cval, 0)
 *           which scipy.ndimage.gaussian_filter1d calls once per filtered axis
 *           (reference call site nd/filters.py:365-378).
 *
 * scipy's NI_Correlate1D arithmetic, restated: with the weights centred at
 * size1 = n/2, an odd-length kernel that is symmetric to DBL_EPSILON gives
 *   out = x[0] w[0];  for j = -size1..-1:  out += (x[j] + x[-j]) * w[j]
 * (anti-symmetric: x[j] - x[-j]); any other kernel
 *   out = x[size2] w[size2];  for j = -size1..size2-1:  out += x[j] * w[j]
 * all in double, cast to the array dtype on store.  `in` and `out` must not
 * overlap.  weights: host pointer, n <= 255 doubles.
 * ---------------------------------------------------------------------- */
int nd_amd_correlate1d(const void *in, void *out, int dtype,
                       const int64_t dims[4],
                       const int64_t in_strides[4], const int64_t out_strides[4],
                       int axis, int nweights, const double *weights,
                       int mode, double cval, void *hip_stream);

/* ------------------------------------------------------------------------
 * Two 1-D correlations in one pass over memory: along axis 2 (y), then along
 * axis 3 (x), the intermediate array rounded to the array dtype exactly as
 * scipy.ndimage.gaussian_filter does between its per-axis passes (it filters
 * `output` in place from the second axis on) -- GaussianFilter(dims=('y','x'))
 * at nd/filters.py:365-378 on x-contiguous planes.
 * Fused form only: float32, x stride 1, both kernels symmetric (to
 * DBL_EPSILON, as NI_Correlate1D tests) and of the same odd length
 * 3, 5, ..., 13 or 17, any border mode except `constant`.  Everything else
 * returns ND_AMD_EUNSUPPORTED and the caller runs two nd_amd_correlate1d
 * passes (nd_amd/kernels.py does).  `in` and `out` must not overlap.
 * ---------------------------------------------------------------------- */
int nd_amd_correlate1d_yx(const void *in, void *out, int dtype,
                          const int64_t dims[4],
                          const int64_t in_strides[4], const int64_t out_strides[4],
                          int nweights, const double *weights_y, const double *weights_x,
                          int mode, void *hip_stream);

/* ------------------------------------------------------------------------
 * Non-local means.
 * Replaces  nd._filters._pixelwise_nlmeans_3d(arr, output, r, f, sigma, h, n_eff)
 *           nd/_filters.pyx:320-420, sole caller nd/filters.py:462.
 *
 *   arr, out      : (N0, N1, N2, nvars) views, element strides given.
 *   patch_mode    : 0 = what the compiled reference does on LP64 platforms:
 *                   `range(-f[i], f[i]+1)` with an unsigned f starts at
 *                   2^32 - f[i], so the patch loops are empty whenever any
 *                   f[i] > 0 and every neighbour gets weight exp(0) = 1
 *                   (nd/_filters.c:3539-3553); 1 = the signed range the
 *                   source text intends (true patch distances).
 *   neff_policy   : find_weight failure (nd/_filters.pyx:310-311):
 *                   0 = self weight 0, as the shipped Cython-0.29 C does;
 *                   1 = report ND_AMD_ENOSOLUTION, as a Cython>=3 build does.
 *   status_dev    : device int32 the kernel sets to 1 on a find_weight
 *                   failure (policy 1); zeroed by the call.  May be NULL for
 *                   n_eff < 0.  The caller reads it after synchronising.
 *   global_N,
 *   tile_off      : for tiled (multi-GPU) use: `arr` is the tile
 *                   [tile_off, tile_off + N) of an array of shape global_N
 *                   that carries its halo rows, `out` likewise; only
 *                   [core_lo, core_hi) of axis `N` is written and reflection
 *                   happens at the GLOBAL edges.  NULL for any of the four means
 *                   the plain call's value: global_N = N, tile_off = 0,
 *                   core = [0, N).
 * ---------------------------------------------------------------------- */
int nd_amd_nlmeans3d(const void *arr, void *out, int dtype,
                     const int64_t N[3], int64_t nvars,
                     const int64_t in_strides[4], const int64_t out_strides[4],
                     const uint32_t r[3], const uint32_t f[3],
                     double sigma, double h, double n_eff,
                     int patch_mode, int neff_policy, int32_t *status_dev,
                     const int64_t global_N[3], const int64_t tile_off[3],
                     const int64_t core_lo[3], const int64_t core_hi[3],
                     void *hip_stream);

/* ------------------------------------------------------------------------
 * nd_amd_omnibus_c2 for data in the reference's own layout, without a
 * transpose: variable v holds element (y, x, t) at
 *     ptr_v[((y * nx + x) * k + t) * date_stride[v]]
 * (order C11, C12re, C12im, C22).  date_stride = 1 for a real (y, x, time)
 * array; 2, with c12im == c12re + 1, for the two halves of an interleaved
 * complex C12 (read once).  Every threshold up to k = 24 dates (float32;
 * float64: 12), whatever the length (lengths that are not a multiple of the
 * 16-byte vector read the staged spans element by element).  Longer series (up to 192 dates, a multiple of 4 -- float64:
 * of 2 --, 16-byte aligned variables) in the sparse regime, alpha >= 0.75:
 * the series is folded out of LDS images instead of being retained, and the
 * search reads a listed pixel's series where it lies.  Everything else
 * returns ND_AMD_EUNSUPPORTED -- transpose with nd_amd_relayout_planar and
 * call nd_amd_omnibus_c2.  Workspace as for nd_amd_omnibus_c2.
 * ---------------------------------------------------------------------- */
int nd_amd_omnibus_c2_pixel_major(const void *c11, const void *c12re, const void *c12im,
                                  const void *c22, int dtype,
                                  int64_t ny, int64_t nx, int64_t k,
                                  const int64_t date_stride[4],
                                  uint32_t n_looks, double alpha, uint8_t *change,
                                  void *z_out, void *p_out,
                                  void *workspace, size_t workspace_bytes, void *hip_stream);

/* ------------------------------------------------------------------------
 * Layout change in front of the hot path.  The reference hands its native
 * code a (y, x, time, variable) view with time (and variable) fastest
 * (nd/change.py:66-67: to_array().transpose('y','x','time','variable'));
 * the kernels above read planar (time, y, x) stacks.  For one variable that
 * is already on the device:
 *     out[t * out_date_stride + p] = in[p * k * in_date_stride + t * in_date_stride]
 * p = flattened (y, x) pixel.  in_date_stride = 1 for a real (y, x, time)
 * array, 2 for the real (in = base) or imaginary (in = base + 1) half of an
 * interleaved complex array (C12).  out_date_stride >= npix.
 * ---------------------------------------------------------------------- */
int nd_amd_relayout_planar(const void *in, void *out, int dtype,
                           int64_t npix, int64_t k, int64_t in_date_stride,
                           int64_t out_date_stride, void *hip_stream);

/* Both halves of an interleaved complex (y, x, time) array in one pass (the C12 term):
 *     out_re[t * s + p] = in[(p * k + t) * 2],  out_im[t * s + p] = in[(p * k + t) * 2 + 1]
 * dtype is that of the real components. */
int nd_amd_relayout_planar_complex(const void *in, void *out_re, void *out_im,
                                   int dtype, int64_t npix, int64_t k,
                                   int64_t out_date_stride, void *hip_stream);

/* The way back (filter outputs handed to a caller that keeps the reference's
 * layout, nd/filters.py:139-176):
 *     out[p * k * out_date_stride + t * out_date_stride] = in[t * in_date_stride + p]
 * out_date_stride = 1 or 2 (one half of an interleaved complex array),
 * in_date_stride >= npix. */
int nd_amd_relayout_pixel_major(const void *in, void *out, int dtype,
                                int64_t npix, int64_t k, int64_t in_date_stride,
                                int64_t out_date_stride, void *hip_stream);

/* ------------------------------------------------------------------------
 * Per-kernel timing with HIP events recorded on the caller's stream
 * (bench.py's roofline figures).  enable(capacity) pre-creates the events;
 * collect() synchronises on them and returns (kernel id, milliseconds) pairs
 * in launch order, then resets.  enable(0) turns timing off.  Launches beyond
 * the capacity are not timed; timing_dropped() says how many that were since
 * the last collect().
 * ---------------------------------------------------------------------- */
int nd_amd_timing_enable(int capacity);
int nd_amd_timing_collect(int32_t *kernel_ids, float *ms, int max_n, int *n_out);
int nd_amd_timing_dropped(void);
/* Restrict the timing to the kernel ids whose bit is set in `id_mask` (bit i = id i; 0 = all, the
 * default after every enable).  An event pair costs a few microseconds of stream time: a benchmark
 * that wants the duration of its dominant kernel inside the timed region times only that one. */
int nd_amd_timing_select(uint64_t id_mask);

/* ------------------------------------------------------------------------
 * Interleaved complex -> real and imaginary arrays of the same contiguous
 * shape: the device side of nd.io.disassemble_complex (nd/io.py:26-69,
 * called at nd/change.py:59 and nd/filters.py:132-134) for a variable that
 * is already in (time, y, x) order.  `in`: n complex values of the real type
 * `dtype` (2 n reals), 16-byte aligned like the outputs.
 * ---------------------------------------------------------------------- */
int nd_amd_split_complex(const void *in, void *out_re, void *out_im, int dtype, int64_t n,
                         void *hip_stream);

/* The inverse (nd.io.assemble_complex, nd/io.py:72-123): two real arrays of n
 * elements -> n interleaved complex values.  16-byte aligned pointers. */
int nd_amd_merge_complex(const void *in_re, const void *in_im, void *out, int dtype, int64_t n,
                         void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* ND_AMD_H */
