// common.h -- shared device helpers for libsassd (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sassd.h"

#define SASSD_WAVE 64

extern int g_sassd_last_hip_error;

static inline int sassd_launch_status()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_sassd_last_hip_error = (int)e;
        return SASSD_EHIP;
    }
    return SASSD_OK;
}

static inline int sassd_hip(hipError_t e)
{
    if (e != hipSuccess) {
        g_sassd_last_hip_error = (int)e;
        return SASSD_EHIP;
    }
    return SASSD_OK;
}

// Opt a kernel into more than 64 KB of dynamic LDS, once per (device, kernel): the attribute is per device, so every
// call site keeps a device bit mask (`static std::atomic<unsigned long long> done{0}`; idempotent if threads race).
#include <atomic>
static inline int sassd_dyn_lds(const void *fn, size_t bytes, std::atomic<unsigned long long> &done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SASSD_EHIP;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return SASSD_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { g_sassd_last_hip_error = (int)e; return SASSD_EHIP; }
    done.fetch_or(bit, std::memory_order_release);
    return SASSD_OK;
}

// Two byte-pattern fills in ONE launch (a workspace that needs an all-ones hash table next to zeroed counters took two
// hipMemsetAsync calls = two __amd_rocclr_fillBufferAligned launches per call: 6 per inference frame in the round-4 trace).
// Both regions 16-byte aligned with sizes that are multiples of 16 (every workspace segment is 256-byte aligned): an ABI
// precondition of the entry points that clear through it (sassd_voxelize, sassd_rulebook_pyramid -- stated in
// include/sassd.h; SASSD_EINVAL before any launch otherwise; hipMemsetAsync, which it replaced, took any pointer).
static __global__ void sassd_fill2_kernel(uint4 *a, size_t na16, unsigned pa, uint4 *b, size_t nb16, unsigned pb)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 va = make_uint4(pa, pa, pa, pa), vb = make_uint4(pb, pb, pb, pb);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na16 + nb16; i += stride) {
        if (i < na16) a[i] = va;
        else b[i - na16] = vb;
    }
}
static inline int sassd_fill2(void *a, size_t abytes, unsigned char pa, void *b, size_t bbytes, unsigned char pb, hipStream_t s)
{
    if (((uintptr_t)a | (uintptr_t)b | abytes | bbytes) & 15) return SASSD_EINVAL;
    const size_t n16 = (abytes + bbytes) / 16;
    if (n16 == 0) return SASSD_OK;
    size_t blocks = (n16 + 4 * 256 - 1) / (4 * 256);             // 4 stores per thread up to 16 MB, then a grid-stride loop
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sassd_fill2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4 *)a, abytes / 16, 0x01010101u * pa,
                       (uint4 *)b, bbytes / 16, 0x01010101u * pb);
    return SASSD_OK;
}

// Compute units of the current device (cached per device; persistent kernels launch one workgroup per CU).
static inline int sassd_num_cus(int *out)
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SASSD_EHIP;
    int n = cached[dev & 63].load(std::memory_order_acquire);
    if (n <= 0) {
        hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || n <= 0) { g_sassd_last_hip_error = (int)e; return SASSD_EHIP; }
        cached[dev & 63].store(n, std::memory_order_release);
    }
    *out = n;
    return SASSD_OK;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline unsigned next_pow2(unsigned x)
{
    unsigned p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---- open-addressing hash: u32 key -> i32 value, linear probing, EMPTY = 0xFFFFFFFF ---------------
#define SASSD_HASH_EMPTY 0xFFFFFFFFu

__device__ __forceinline__ unsigned hash_u32(unsigned k)
{
    k ^= k >> 16;
    k *= 0x7feb352du;
    k ^= k >> 15;
    k *= 0x846ca68bu;
    k ^= k >> 16;
    return k;
}

// returns slot index of `key` (inserting it if absent) or -1 if the table is full
__device__ __forceinline__ int hash_insert(unsigned *keys, unsigned mask, unsigned key)
{
    unsigned h = hash_u32(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        unsigned prev = atomicCAS(&keys[h], SASSD_HASH_EMPTY, key);
        if (prev == SASSD_HASH_EMPTY || prev == key) return (int)h;
        h = (h + 1) & mask;
    }
    return -1;
}

__device__ __forceinline__ int hash_find(const unsigned *keys, unsigned mask, unsigned key)
{
    unsigned h = hash_u32(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        unsigned k = keys[h];
        if (k == key) return (int)h;
        if (k == SASSD_HASH_EMPTY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

// ---- the rulebook's coordinate table: INTERLEAVED entries {key u32, value i32} (round 4) ------------------------
// A lookup is one 8-byte load per probe; with separate key / value arrays (rounds 1-3) every hit paid a second,
// dependent load from another cache line -- the level-0 submanifold table and the first strided table (26 / 27 lookups
// per row) are chains of such round trips.  EMPTY key = 0xFFFFFFFF (the table is cleared with 0xFF bytes).
struct HashView {
    uint2 *ent;                // [h] {key, value}, h = pow2 >= 2 * cap_rows
    unsigned mask;
};

// slot of `key` (claiming it if absent) or -1 if the table is full
__device__ __forceinline__ int hash2_insert(uint2 *ent, unsigned mask, unsigned key)
{
    unsigned h = hash_u32(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        unsigned prev = atomicCAS(&ent[h].x, SASSD_HASH_EMPTY, key);
        if (prev == SASSD_HASH_EMPTY || prev == key) return (int)h;
        h = (h + 1) & mask;
    }
    return -1;
}

// value stored for `key`, or -1
__device__ __forceinline__ int hash2_lookup(const uint2 *ent, unsigned mask, unsigned key)
{
    unsigned h = hash_u32(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const uint2 e = ent[h];
        if (e.x == key) return (int)e.y;
        if (e.x == SASSD_HASH_EMPTY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

static inline unsigned hash_cap(int cap_rows)
{
    unsigned h = next_pow2((unsigned)(cap_rows > 0 ? cap_rows : 1) * 2u);
    return h < 1024 ? 1024 : h;
}

// host-side view of a table buffer of sassd_hash_bytes(cap_rows) = 8 * h bytes
static inline HashView hash_view(const void *table, int cap_rows)
{
    HashView v;
    v.ent = (uint2 *)table;
    v.mask = hash_cap(cap_rows) - 1;
    return v;
}

// ---- block-wide exclusive scan of one int per thread (blockDim.x multiple of 64, <= 1024) ---------
// `wsum` is shared scratch of >= 17 ints. Returns the exclusive prefix; *total = block sum.
__device__ __forceinline__ int block_exclusive_scan(int v, int *wsum, int *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int s = (lane < nw) ? wsum[lane] : 0;
        int si = s;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            int t = __shfl_up(si, o, 64);
            if (lane >= o) si += t;
        }
        if (lane < nw) wsum[lane] = si - s;      // exclusive prefix of wave sums
        if (lane == nw - 1) wsum[16] = si;       // block total
    }
    __syncthreads();
    int res = wsum[wid] + inc - v;
    *total = wsum[16];
    __syncthreads();                              // wsum may be reused by the caller
    return res;
}
