// Tone mapping of the radiance accumulator on the device (reference imgutils.py:60-80, the
// flare-free path): 8-bit RGB out of the float sums without bringing 16 B per pixel to the host.
// Arithmetic is binary64 like the numpy pipeline it replaces.  `img` is in ray order; `order`
// gives (row, col) of every ray; `base` (optional) = earlier sums to add (--append).
typedef unsigned int uint;

static __device__ __forceinline__ double clr_tm_clean(double v) {
    // numpy.nan_to_num: NaN -> 0, +-inf -> +-largest finite
    if (v != v) return 0.0;
    if (v > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (v < -1.7976931348623157e308) return -1.7976931348623157e308;
    return v;
}

static __device__ __forceinline__ void clr_tm_pixel(const float4 *img, const double *base, int k, double rgb[3]) {
    const float4 v = img[k];
    rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
    if (base) { rgb[0] += base[3*k]; rgb[1] += base[3*k + 1]; rgb[2] += base[3*k + 2]; }
    for (int c = 0; c < 3; ++c) rgb[c] = clr_tm_clean(rgb[c]);
}

// per block: minimum and sum of the lightness (r + g + b) / 3
extern "C" __global__ void __launch_bounds__(256)
clr_tm_reduce(const float4 * __restrict__ img, const double * __restrict__ base, const int n_pixels,
              double * __restrict__ block_min, double * __restrict__ block_sum)
{
    __shared__ double s_min[256], s_sum[256];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double lo = 1.7976931348623157e308, sum = 0.0;
    if (k < n_pixels) {
        double rgb[3];
        clr_tm_pixel(img, base, k, rgb);
        lo = sum = (rgb[0] + rgb[1] + rgb[2]) / 3.0;
    }
    s_min[threadIdx.x] = lo;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) {
            s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + step]);
            s_sum[threadIdx.x] += s_sum[threadIdx.x + step];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { block_min[blockIdx.x] = s_min[0]; block_sum[blockIdx.x] = s_sum[0]; }
}

// out[(row * width + col) * 3 + c] = 8-bit value
extern "C" __global__ void __launch_bounds__(256)
clr_tm_encode(const float4 * __restrict__ img, const double * __restrict__ base,
              const int2 * __restrict__ order, const int n_pixels, const int width,
              const double floor_value, const double scale, const double inv_gamma,
              unsigned char * __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    double rgb[3];
    clr_tm_pixel(img, base, k, rgb);
    const int2 rc = order[k];
    unsigned char *dst = out + ((size_t)rc.x * width + rc.y) * 3;
    for (int c = 0; c < 3; ++c) {
        double v = (rgb[c] - floor_value) * scale;
        v = v > 0.0 ? v : 0.0;
        v = pow(v, inv_gamma);
        v = v < 1.0 ? v : 1.0;
        dst[c] = (unsigned char)(v * 255.0);
    }
}

// ---- flares (reference imgutils.py:121-148): over-exposed values bleed into their
// neighbourhood through a 9-tap gaussian (both axes, mirrored at the border) and then through a
// 198-tap exponential tail along the rows axis (zero outside the frame).  The work planes are
// (H, W, 3) binary64 in image order; the filters follow scipy.ndimage's correlate1d term by term
// (symmetric form: centre first, then pairs from the outermost inwards; general form: taps in
// ascending order), so that the sums round as on the host.  Built with -ffp-contract=off.

// value after scaling and gamma, not clamped above, written in image order
extern "C" __global__ void __launch_bounds__(256)
clr_tm_linear(const float4 * __restrict__ img, const double * __restrict__ base,
              const int2 * __restrict__ order, const int n_pixels, const int width,
              const double floor_value, const double scale, const double inv_gamma,
              double * __restrict__ plane)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    double rgb[3];
    clr_tm_pixel(img, base, k, rgb);
    const int2 rc = order[k];
    double *dst = plane + ((size_t)rc.x * width + rc.y) * 3;
    for (int c = 0; c < 3; ++c) {
        double v = (rgb[c] - floor_value) * scale;
        v = v > 0.0 ? v : 0.0;
        dst[c] = pow(v, inv_gamma);
    }
}

// numpy's clip(v, 0, limit) and the part of v above it, v - clip(v, 0, limit); limit < 0: v itself
static __device__ __forceinline__ double clr_tm_kept(double v, double limit) {
    return v < 0.0 ? 0.0 : (v > limit ? limit : v);
}
static __device__ __forceinline__ double clr_tm_excess(double v, double limit) {
    return limit < 0.0 ? v : v - clr_tm_kept(v, limit);
}

// one 1-D correlation along an axis of the plane.  Element e lies at position
// (e / stride) % length of its line; its neighbours are `stride` elements apart.
// taps = 2 * reach + 1 mirrored (symmetric != 0), else `taps` weights starting `reach` before e.
extern "C" __global__ void __launch_bounds__(256)
clr_tm_correlate(const double * __restrict__ src, double * __restrict__ dst, const long n_elements,
                 const long stride, const int length, const double * __restrict__ weights,
                 const int taps, const int reach, const int symmetric, const double limit)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elements) return;
    const int at = (int)((e / stride) % length);
    const double *line = src + (e - at * stride);
    double sum;
    if (symmetric) {
        const double *centre = weights + reach;
        sum = clr_tm_excess(line[at * stride], limit) * centre[0];
        for (int j = -reach; j < 0; ++j) {
            int lo = at + j, hi = at - j;
            // mirror about the edges of the samples: (d c b a | a b c d | d c b a)
            while (lo < 0 || lo >= length) lo = lo < 0 ? -lo - 1 : 2 * length - 1 - lo;
            while (hi < 0 || hi >= length) hi = hi < 0 ? -hi - 1 : 2 * length - 1 - hi;
            sum += (clr_tm_excess(line[lo * stride], limit) + clr_tm_excess(line[hi * stride], limit)) * centre[j];
        }
    } else {
        sum = 0.0;
        for (int m = 0; m < taps; ++m) {
            const int p = at - reach + m;
            const double v = (p < 0 || p >= length) ? 0.0 : clr_tm_excess(line[p * stride], limit);
            sum += v * weights[m];
        }
    }
    dst[e] = sum;
}

// plane = clip(plane, 0, limit) + bleed
extern "C" __global__ void __launch_bounds__(256)
clr_tm_merge(double * __restrict__ plane, const double * __restrict__ bleed, const long n_elements,
             const double limit)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elements) return;
    plane[e] = clr_tm_kept(plane[e], limit) + bleed[e];
}

// 8 bit of clip(clip(plane, 0, limit) + bleed, 0, 1)
extern "C" __global__ void __launch_bounds__(256)
clr_tm_encode_plane(const double * __restrict__ plane, const double * __restrict__ bleed,
                    const long n_elements, const double limit, unsigned char * __restrict__ out)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elements) return;
    double v = clr_tm_kept(plane[e], limit) + bleed[e];
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    out[e] = (unsigned char)(v * 255.0);
}
