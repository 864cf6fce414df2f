// spconv.hip -- output-stationary sparse 3-D convolution forward with LDS-staged gather / scatter.
//
// Replaces spconv v1.0 `indice_conv_fp32` (27 x {gather kernel, cuBLAS sgemm, scatter-add kernel} per layer,
// ~80 launches) + nn.BatchNorm1d + nn.ReLU (mmdet/models/necks/cmn.py:138-173,208-212) by ONE launch per layer.
//
// Work decomposition (wave = 64 lanes, CDNA4):
//   workgroup = tile of 64 output rows, NT = Cout/16 waves; wave w owns output channels [16w, 16w+16).
//   1. the tile's rulebook records nbr[64][27] are copied to LDS (one coalesced 6.9 KB read);
//   2. per kernel offset k a wave ballots "row has a neighbour at k" and compacts (row, in_row) pairs into an
//      LDS list -- only real pairs are multiplied (dense 27-offset evaluation would waste ~3x the FLOPs);
//   3. pairs are processed 16 at a time: A = 16 gathered input rows (each lane loads Cin/4 contiguous floats of
//      its row straight from HBM/L2), B = W[k][:, 16 couts] fragment (pre-packed, 16 B/lane loads);
//      Cin/4 v_mfma_f32_16x16x4_f32 steps (exact fp32, the fp32 MFMA rate equals the fp32 VALU rate but needs one
//      VGPR per operand);  results are scatter-ACCUMULATED into the tile's fp32 accumulator in LDS
//      (each wave owns its 16 columns -> plain read-modify-write, deterministic summation order);
//   4. epilogue: LDS accumulator -> scale/shift/ReLU -> coalesced float4 stores.
// Roofline: HBM/L2 bound (<= 16 FLOP/B); algorithmic bytes B_gs = 4P(Cin+Cout) + 8P + 4K*Cin*Cout + 4*Nout*Cout.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 64;      // output rows per workgroup
constexpr int kK = 27;

template <int CIN, int COUT>
struct SpShape {
    static constexpr int KS = CIN / 4;            // floats per lane per operand = MFMA steps
    static constexpr int NT = COUT / 16;          // waves per workgroup
    static constexpr int LDA = COUT + 4;          // accumulator row stride (floats), keeps float4 alignment
};

template <int N>
__device__ __forceinline__ void load_vec(const float *__restrict__ p, float (&v)[N])
{
    if constexpr (N >= 4) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const float4 t = ((const float4 *)p)[i];
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}

// w [K][CIN][COUT] -> packed [K][NT][64 lanes][KS]:  lane (q = l>>4, m = l&15) holds W[k][q*KS + kk][nt*16 + m]
template <int CIN, int COUT>
__global__ void pack_weight_kernel(const float *__restrict__ w, int K, float *__restrict__ packed)
{
    using S = SpShape<CIN, COUT>;
    const int total = K * S::NT * 64 * S::KS;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kk = i % S::KS;
    const int lane = (i / S::KS) % 64;
    const int nt = (i / (S::KS * 64)) % S::NT;
    const int k = i / (S::KS * 64 * S::NT);
    const int q = lane >> 4, m = lane & 15;
    packed[i] = w[((size_t)k * CIN + q * S::KS + kk) * COUT + nt * 16 + m];
}

template <int CIN, int COUT>
__global__ void __launch_bounds__((COUT / 16) * 64)
spconv_fwd_kernel(const float *__restrict__ x, const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_ptr,
                  int cap, const float *__restrict__ wp, int K, const float *__restrict__ scale,
                  const float *__restrict__ shift, int relu, float *__restrict__ y)
{
    using S = SpShape<CIN, COUT>;
    constexpr int KS = S::KS, NT = S::NT, LDA = S::LDA;
    __shared__ __attribute__((aligned(16))) float acc_s[kTile * LDA];
    __shared__ int nbr_s[kTile * kK];
    __shared__ int list_idx[kK * kTile];
    __shared__ unsigned char list_row[kK * kTile];
    __shared__ int cnt_s[kK];

    const int n = min(*n_ptr, cap);
    const int r0 = blockIdx.x * kTile;
    if (r0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = min(kTile, n - r0);

    // ---- 1. stage rulebook records + zero the accumulator ----------------------------------------
    if (nbr) {
        const int32_t *src = nbr + (size_t)r0 * kK;
        for (int i = tid; i < rows * kK; i += NT * 64) nbr_s[i] = src[i];
    }
    for (int i = tid; i < kTile * LDA; i += NT * 64) acc_s[i] = 0.f;
    __syncthreads();

    // ---- 2. per-offset compaction (wave w handles offsets w, w+NT, ...) --------------------------
    for (int k = wave; k < K; k += NT) {
        int v = -1;
        if (lane < rows) v = nbr ? nbr_s[lane * kK + k] : (r0 + lane);
        const unsigned long long m = __ballot(v >= 0);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (v >= 0) { list_idx[k * kTile + rank] = v; list_row[k * kTile + rank] = (unsigned char)lane; }
        if (lane == 0) cnt_s[k] = __popcll(m);
    }
    __syncthreads();

    // ---- 3. gather -> MFMA -> LDS scatter-accumulate ---------------------------------------------
    const int q = lane >> 4, m16 = lane & 15;
    const float *wbase = wp + ((size_t)wave * 64 + lane) * KS;          // + k * NT*64*KS
    for (int k = 0; k < K; ++k) {
        const int c = cnt_s[k];
        if (c == 0) continue;
        float bf[KS];
        load_vec<KS>(wbase + (size_t)k * NT * 64 * KS, bf);
        for (int j0 = 0; j0 < c; j0 += 16) {
            const int p = j0 + m16;
            float af[KS];
            if (p < c) {
                const int idx = list_idx[k * kTile + p];
                load_vec<KS>(x + (size_t)idx * CIN + q * KS, af);
            } else {
#pragma unroll
                for (int i = 0; i < KS; ++i) af[i] = 0.f;
            }
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; kk += 2) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bf[kk], d0, 0, 0, 0);
                if (kk + 1 < KS) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk + 1], bf[kk + 1], d1, 0, 0, 0);
            }
            // D[row = q*4 + reg][col = m16]  -> pair j0 + q*4 + reg, channel wave*16 + m16
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int pr = j0 + q * 4 + reg;
                if (pr < c) {
                    const int rl = list_row[k * kTile + pr];
                    acc_s[rl * LDA + wave * 16 + m16] += d0[reg] + d1[reg];
                }
            }
        }
    }
    __syncthreads();

    // ---- 4. epilogue -----------------------------------------------------------------------------
    constexpr int C4 = COUT / 4;
    for (int i = tid; i < rows * C4; i += NT * 64) {
        const int r = i / C4, c4 = i - r * C4;
        float4 v = *(const float4 *)&acc_s[r * LDA + c4 * 4];
        if (scale) {
            const float4 s = ((const float4 *)scale)[c4];
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        }
        if (shift) {
            const float4 s = ((const float4 *)shift)[c4];
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        ((float4 *)(y + (size_t)(r0 + r) * COUT))[c4] = v;
    }
}

template <int CIN, int COUT>
int launch_fwd(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp, int K,
               const float *scale, const float *shift, int relu, float *y, hipStream_t stream)
{
    using S = SpShape<CIN, COUT>;
    hipLaunchKernelGGL((spconv_fwd_kernel<CIN, COUT>), dim3(cdiv(cap, kTile)), dim3(S::NT * 64), 0, stream, x, nbr,
                       n_ptr, cap, wp, K, scale, shift, relu, y);
    return sassd_launch_status();
}

template <int CIN, int COUT>
int launch_pack(const float *w, int K, float *packed, hipStream_t stream)
{
    using S = SpShape<CIN, COUT>;
    const int total = K * S::NT * 64 * S::KS;
    hipLaunchKernelGGL((pack_weight_kernel<CIN, COUT>), dim3(cdiv(total, 256)), dim3(256), 0, stream, w, K, packed);
    return sassd_launch_status();
}

#define SP_DISPATCH(FN, ...)                                                     \
    if (Cin == 4 && Cout == 16) return FN<4, 16>(__VA_ARGS__);                   \
    if (Cin == 16 && Cout == 16) return FN<16, 16>(__VA_ARGS__);                 \
    if (Cin == 16 && Cout == 32) return FN<16, 32>(__VA_ARGS__);                 \
    if (Cin == 32 && Cout == 32) return FN<32, 32>(__VA_ARGS__);                 \
    if (Cin == 32 && Cout == 64) return FN<32, 64>(__VA_ARGS__);                 \
    if (Cin == 64 && Cout == 64) return FN<64, 64>(__VA_ARGS__);                 \
    return SASSD_EINVAL;

__global__ void densify_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx,
                               const int32_t *__restrict__ n_ptr, int cap, int C, int D, int H, int W, int order,
                               float *__restrict__ out)
{
    const int n = min(*n_ptr, cap);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    // consecutive threads -> consecutive ROWS for a fixed channel: writes of neighbouring voxels (ascending x)
    // land in the same cache line of one channel plane
    const int c = t / n, row = t - c * n;
    const int4 p = ((const int4 *)idx)[row];
    const int ch = order ? (p.y * C + c) : (c * D + p.y);
    out[(((size_t)p.x * C * D + ch) * H + p.z) * W + p.w] = feats[(size_t)row * C + c];
}

}  // namespace

extern "C" size_t sassd_spconv_packed_floats(int K, int Cin, int Cout) { return (size_t)K * Cin * Cout; }

extern "C" int sassd_spconv_pack_weight(const float *w, int K, int Cin, int Cout, float *packed, void *stream_)
{
    if (!w || !packed || K < 1 || K > kK) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    SP_DISPATCH(launch_pack, w, K, packed, stream)
}

extern "C" int sassd_spconv_fwd(const float *x, const int32_t *nbr, const int32_t *n_out_ptr, int cap_out,
                                const float *w_packed, int K, int Cin, int Cout, const float *scale,
                                const float *shift, int relu, float *y, void *stream_)
{
    if (!x || !n_out_ptr || !w_packed || !y || cap_out <= 0) return SASSD_EINVAL;
    if (nbr ? (K != kK) : (K != 1)) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    SP_DISPATCH(launch_fwd, x, nbr, n_out_ptr, cap_out, w_packed, K, scale, shift, relu, y, stream)
}

extern "C" int sassd_densify(const float *feats, const int32_t *indices, const int32_t *n_ptr, int cap, int C,
                             int D, int H, int W, int batch_size, int channel_order, float *out, void *stream_)
{
    if (!feats || !indices || !n_ptr || !out || cap <= 0) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = sassd_hip(hipMemsetAsync(out, 0, (size_t)batch_size * C * D * H * W * sizeof(float), stream)))) return rc;
    hipLaunchKernelGGL(densify_kernel, dim3(cdiv(cap * C, 256)), dim3(256), 0, stream, feats, indices, n_ptr, cap, C, D,
                       H, W, channel_order, out);
    return sassd_launch_status();
}
