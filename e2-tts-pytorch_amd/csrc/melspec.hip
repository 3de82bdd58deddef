// Fused STFT-magnitude -> mel filterbank -> log for MelSpec (e2_tts.py:248-290; torchaudio MelSpectrogram semantics
// restated in SURVEY.md Appendix A.8): reflect padding (center=True), periodic Hann window, 1024-point FFT, |.|,
// (513 x n_mels) htk filterbank, log(clamp(min=1e-5)).  The reference runs this as torch.stft (cuFFT-class library
// call) + abs + matmul + transpose + clamp + log, i.e. the (B, 513, frames) complex spectrum and magnitude make
// four HBM round trips; here a workgroup keeps 4 frames in LDS from the raw samples to the log-mel row.
// HBM-bound: algorithmic bytes = 4*nw (samples, read once; the 4x frame overlap is served from L2) + 4*n_mels*frames.
#include "e2k_device.h"
#include "plan.h"
#include "../../include/e2k.h"

using namespace e2k;

namespace {

constexpr int NFFT = 1024, LOGN = 10, NBIN = NFFT / 2 + 1, FR = 4;

__device__ __forceinline__ int bitrev10(int x) {
    int r = 0;
#pragma unroll
    for (int i = 0; i < LOGN; ++i) r |= ((x >> i) & 1) << (LOGN - 1 - i);
    return r;
}

struct MelArgs {
    const float* wave; long nw; const float* window; const float* fb; const float* twc; const float* tws;
    float* out; int B, frames, hop, n_mels;
    const int* lens; float pad_value;        // ragged batch: valid samples per row (nullptr = every row has nw), fill of the frames past a row's end
    const int* bands;                        // optional [n_mels][2]: bins [lo, hi) outside of which column m of fb is zero
};

__global__ __launch_bounds__(256) void melspec_kernel(MelArgs p) {
    __shared__ float re[FR][NFFT];
    __shared__ float im[FR][NFFT];
    __shared__ float tc[NFFT / 2], ts[NFFT / 2];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * FR;
    const float* x = p.wave + (long)b * p.nw;
    // ragged batch: this row holds n valid samples; the reflection at its end and its frame count follow n, exactly as if
    // the row had been transformed on its own (the reference's dataset does one clip at a time, trainer.py:101-131)
    const long n = p.lens ? max(2L, min((long)p.lens[b], p.nw)) : p.nw;
    const int nframes = p.lens ? (int)(1 + n / p.hop) : p.frames;
    for (int i = tid; i < NFFT / 2; i += 256) { tc[i] = p.twc[i]; ts[i] = p.tws[i]; }
    // load + reflect pad + window, stored bit-reversed
    for (int fr = 0; fr < FR; ++fr) {
        const int f = f0 + fr;
        for (int i = tid; i < NFFT; i += 256) {
            float v = 0.f;
            if (f < nframes) {
                long j = (long)f * p.hop + i - NFFT / 2;
                if (j < 0) j = -j;
                if (j >= n) j = 2 * (n - 1) - j;
                j = max(0L, min(j, n - 1));          // (only rows shorter than n_fft / 2 + 1 get here: torch.stft rejects those)
                v = x[j] * p.window[i];
            }
            const int r = bitrev10(i);
            re[fr][r] = v;
            im[fr][r] = 0.f;
        }
    }
    __syncthreads();
    for (int s = 1; s <= LOGN; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = NFFT >> s;
        for (int j = tid; j < NFFT / 2; j += 256) {
            const int grp = j >> (s - 1), pos = j & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            const float c = tc[pos * tstep], sn = ts[pos * tstep];      // W = c - i*sn
#pragma unroll
            for (int fr = 0; fr < FR; ++fr) {
                const float xr = re[fr][i1], xi = im[fr][i1];
                const float tr = xr * c + xi * sn;
                const float ti = xi * c - xr * sn;
                const float ur = re[fr][i0], ui = im[fr][i0];
                re[fr][i0] = ur + tr; im[fr][i0] = ui + ti;
                re[fr][i1] = ur - tr; im[fr][i1] = ui - ti;
            }
        }
        __syncthreads();
    }
    // magnitude into re[fr][0..512]
    for (int k = tid; k < NBIN; k += 256) {
#pragma unroll
        for (int fr = 0; fr < FR; ++fr) {
            const float a = re[fr][k], c = im[fr][k];
            re[fr][k] = sqrtf(a * a + c * c);
        }
    }
    __syncthreads();
    if (p.bands) {
        // the htk triangles overlap only their neighbours: a bin feeds at most two bands, band m reads bins [lo, hi) only
        // (513 x 100 dense = 51 k multiply-adds per frame, of which 1 k are non-zero).  One (band, frame) pair per thread.
        for (int idx = tid; idx < p.n_mels * FR; idx += 256) {
            const int m = idx % p.n_mels, fr = idx / p.n_mels, f = f0 + fr;
            if (f >= p.frames) continue;
            const int lo = max(0, p.bands[2 * m]), hi = min(NBIN, p.bands[2 * m + 1]);
            float acc = 0.f;
            for (int k = lo; k < hi; ++k) acc = fmaf(re[fr][k], p.fb[(long)k * p.n_mels + m], acc);
            p.out[((long)b * p.n_mels + m) * p.frames + f] = f < nframes ? logf(fmaxf(acc, 1e-5f)) : p.pad_value;
        }
        return;
    }
    if (tid < p.n_mels) {
        float acc[FR];
#pragma unroll
        for (int fr = 0; fr < FR; ++fr) acc[fr] = 0.f;
        for (int k = 0; k < NBIN; ++k) {
            const float w = p.fb[(long)k * p.n_mels + tid];
#pragma unroll
            for (int fr = 0; fr < FR; ++fr) acc[fr] = fmaf(re[fr][k], w, acc[fr]);
        }
#pragma unroll
        for (int fr = 0; fr < FR; ++fr) {
            const int f = f0 + fr;
            if (f < p.frames) p.out[((long)b * p.n_mels + tid) * p.frames + f] = f < nframes ? logf(fmaxf(acc[fr], 1e-5f)) : p.pad_value;
        }
    }
}

// Polyphase sinc resampler of the dataset path (trainer.py:116-118: torchaudio.transforms.Resample(sample_rate, target_sample_rate) per
// clip; torchaudio's _apply_sinc_resample_kernel restated): with orig / new the two rates divided by their gcd, output sample
// j * new + ph of a row is sum_k kernel[ph][k] * xpad[j * orig + k], xpad = the row with `width` zeros in front and zeros behind it
// (a strided conv1d with `new` output channels).  One lane per output sample; a workgroup's 256 consecutive outputs read a window of
// ~256 * orig / new + taps input samples, which L2 / L1 serve (HBM-bound: 4 B in + 4 B out per sample, the taps are MACs on cached data).
// Rows are independent: a zero-padded ragged batch gives each row what it would get alone; outputs past ceil(len * new / orig) are zero.
struct ResampleArgs {
    const float* x; long ldx; long n_in; const int* lens; const float* kernel; float* out; long ldo; long n_out;
    int orig, nw, taps, width;
};

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs p) {
    const int b = blockIdx.y;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.n_out) return;
    const long len = p.lens ? min((long)p.lens[b], p.n_in) : p.n_in;
    const long valid = (len * p.nw + p.orig - 1) / p.orig;          // ceil(new * length / orig)
    float acc = 0.f;
    if (o < valid) {
        const long j = o / p.nw;
        const int ph = (int)(o - j * p.nw);
        const float* kr = p.kernel + (long)ph * p.taps;
        const float* xr = p.x + (long)b * p.ldx;
        const long i0 = j * p.orig - p.width;                        // input index of tap 0
        const int k0 = (int)max(0L, -i0), k1 = (int)min((long)p.taps, len - i0);
        for (int k = k0; k < k1; ++k) acc = fmaf(kr[k], xr[i0 + k], acc);
    }
    p.out[(long)b * p.ldo + o] = acc;
}

}  // namespace

static int resample_impl(const float* x, int64_t ldx, int64_t n_in, const int32_t* lens, const float* kernel, float* out, int64_t ldo,
                         int64_t n_out, int B, int orig, int nw, int taps, int width, void* stream) {
    if (B <= 0 || n_out <= 0) return 0;
    if (!x || !kernel || !out) return E2K_ERR_ARG;
    if (orig <= 0 || nw <= 0 || taps <= 0 || width < 0 || n_in <= 0 || ldx < n_in || ldo < n_out) return E2K_ERR_SHAPE;
    ResampleArgs a{x, (long)ldx, (long)n_in, lens, kernel, out, (long)ldo, (long)n_out, orig, nw, taps, width};
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, a);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int melspec_impl(const float* wave, int64_t nw, const float* window, const float* fb, const float* twc,
                           const float* tws, float* out, int B, int n_fft, int hop, int n_mels, const int32_t* bands, void* stream) {
    if (B <= 0 || nw <= 0) return 0;
    if (n_fft != NFFT || n_mels > 256 || n_mels <= 0 || hop <= 0 || nw <= NFFT / 2) return E2K_ERR_SHAPE;
    if (!wave || !window || !fb || !twc || !tws || !out) return E2K_ERR_ARG;
    MelArgs a{wave, (long)nw, window, fb, twc, tws, out, B, (int)(1 + nw / hop), hop, n_mels, nullptr, 0.f, bands};
    hipLaunchKernelGGL(melspec_kernel, dim3((a.frames + FR - 1) / FR, B), dim3(256), 0, (hipStream_t)stream, a);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int melspec_ragged_impl(const float* wave, int64_t nw, const int32_t* lens, const float* window, const float* fb,
                                  const float* twc, const float* tws, float* out, float pad_value, int B, int n_fft, int hop,
                                  int n_mels, const int32_t* bands, void* stream) {
    if (B <= 0 || nw <= 0) return 0;
    if (n_fft != NFFT || n_mels > 256 || n_mels <= 0 || hop <= 0 || nw <= NFFT / 2) return E2K_ERR_SHAPE;
    if (!wave || !lens || !window || !fb || !twc || !tws || !out) return E2K_ERR_ARG;
    MelArgs a{wave, (long)nw, window, fb, twc, tws, out, B, (int)(1 + nw / hop), hop, n_mels, lens, pad_value, bands};
    hipLaunchKernelGGL(melspec_kernel, dim3((a.frames + FR - 1) / FR, B), dim3(256), 0, (hipStream_t)stream, a);
    E2K_CHECK_LAUNCH();
    return 0;
}

// ---- C ABI: every compute entry point goes through e2k::dispatch (plan.h) so that a launch plan can record it

extern "C" int e2k_melspec(const float* wave, int64_t nw, const float* window, const float* fb, const float* twc,
                           const float* tws, float* out, int B, int n_fft, int hop, int n_mels, const int32_t* bands, void* stream) {
    return e2k::dispatch("melspec", melspec_impl, wave, nw, window, fb, twc, tws, out, B, n_fft, hop, n_mels, bands, stream);
}

extern "C" int e2k_melspec_ragged(const float* wave, int64_t nw, const int32_t* lens, const float* window, const float* fb,
                                  const float* twc, const float* tws, float* out, float pad_value, int B, int n_fft, int hop,
                                  int n_mels, const int32_t* bands, void* stream) {
    return e2k::dispatch("melspec_ragged", melspec_ragged_impl, wave, nw, lens, window, fb, twc, tws, out, pad_value, B, n_fft, hop, n_mels, bands, stream);
}

extern "C" int e2k_resample_sinc(const float* x, int64_t ldx, int64_t n_in, const int32_t* lens, const float* kernel, float* out, int64_t ldo,
                                 int64_t n_out, int B, int orig, int nw, int taps, int width, void* stream) {
    return e2k::dispatch("resample_sinc", resample_impl, x, ldx, n_in, lens, kernel, out, ldo, n_out, B, orig, nw, taps, width, stream);
}
