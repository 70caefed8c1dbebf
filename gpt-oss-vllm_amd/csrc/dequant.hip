// Standalone NF4 -> fp16 dequantizer and the checkpoint-order -> tile-order repack.
// Both are pure byte movers bound by HBM: 0.515625 B read + 2 B written per weight for the
// dequantizer, one read and one write of every byte for the repack.
#include "common.h"

namespace m355 {

// A work item turns 4 packed bytes (8 weights) into 16 bytes of fp16.  A wave takes 256 consecutive items per step
// (1 KiB in, 4 KiB out): lane l handles items l, l + 64, l + 128, l + 192 of them, so every load instruction reads 256
// contiguous bytes and every store instruction writes 1 KiB contiguous.  The code book is looked up per byte in LDS as
// a float2 (low-nibble code, high-nibble code) from a table with 256-byte rows and one replica per lane of a 32-lane
// group (64 KiB, conflict-free ds_read_b64; the address of a lookup is one v_perm_b32: the byte dropped into bits 8..15
// above the lane's replica offset, which needs the table at LDS address 0 -- it is the kernel's only LDS).  Products are
// taken in fp32 and rounded to nearest even, which reproduces the reference bit for bit.  The block of an item is
// item >> shift when blocksize / 8 is a power of two (the checkpoint's 64 and 128 are), else a 64-bit division.
constexpr int DQ_THREADS = 1024;   // two workgroups per CU share the LDS: 32 waves per CU keep enough loads in flight
__device__ const float nf4_code[16] = M355_NF4_TABLE;

__global__ __launch_bounds__(DQ_THREADS) void nf4_dequant_kernel(const uint32_t* __restrict__ q,
                                                                 const float* __restrict__ absmax, int64_t n_items,
                                                                 int items_per_block, int shift, u32x4_t* __restrict__ out) {
    typedef float __attribute__((ext_vector_type(2))) f2_t;
    __shared__ __attribute__((aligned(16))) f2_t lut[256 * 32];
    {
        // thread t writes replica t & 31 of byte values (t >> 5) + 32 i: a 32-lane group touches every bank pair once
        const int r = threadIdx.x & 31, b0 = threadIdx.x >> 5;
#pragma unroll
        for (int i = 0; i < 256 * 32 / DQ_THREADS; i++) {
            const int b = b0 + (DQ_THREADS / 32) * i;
            lut[b * 32 + r] = f2_t{nf4_code[b & 15], nf4_code[b >> 4]};
        }
    }
    __syncthreads();
    typedef const f2_t __attribute__((address_space(3))) lds_f2_t;
    const int lane = threadIdx.x & 63;
    const uint32_t lbase = (uint32_t)(size_t)(lds_f2_t*)lut + (lane & 31) * 8;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (DQ_THREADS / 64) + (threadIdx.x >> 6)));
    const int64_t n_waves = (int64_t)gridDim.x * (DQ_THREADS / 64);
    constexpr int U = 4;
    for (int64_t base = wave0 * (64 * U); base < n_items; base += n_waves * (64 * U)) {
        uint32_t pk[U];
        float am[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t it = base + u * 64 + lane;
            const bool ok = it < n_items;
            pk[u] = ok ? __builtin_nontemporal_load(q + it) : 0u;
            am[u] = ok ? absmax[shift >= 0 ? (it >> shift) : (it / items_per_block)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t it = base + u * 64 + lane;
            uint32_t a0 = __builtin_amdgcn_perm(pk[u], lbase, 0x03020400u), a1 = __builtin_amdgcn_perm(pk[u], lbase, 0x03020500u);
            uint32_t a2 = __builtin_amdgcn_perm(pk[u], lbase, 0x03020600u), a3 = __builtin_amdgcn_perm(pk[u], lbase, 0x03020700u);
            const f2_t c0 = *reinterpret_cast<lds_f2_t*>(a0), c1 = *reinterpret_cast<lds_f2_t*>(a1);
            const f2_t c2 = *reinterpret_cast<lds_f2_t*>(a2), c3 = *reinterpret_cast<lds_f2_t*>(a3);
            const half2_t h0 = {(half_t)(c0.x * am[u]), (half_t)(c0.y * am[u])}, h1 = {(half_t)(c1.x * am[u]), (half_t)(c1.y * am[u])};
            const half2_t h2 = {(half_t)(c2.x * am[u]), (half_t)(c2.y * am[u])}, h3 = {(half_t)(c3.x * am[u]), (half_t)(c3.y * am[u])};
            const u32x4_t o = {__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1), __builtin_bit_cast(uint32_t, h2),
                               __builtin_bit_cast(uint32_t, h3)};
            if (it < n_items) __builtin_nontemporal_store(o, out + it);
        }
    }
}

// Row of the checkpoint matrix that becomes row `row` of the kernels' order: as is, or (row_order 1) the even
// checkpoint rows first, then the odd ones -- gate/up interleaved in the checkpoint, gate rows then up rows here.
__device__ __forceinline__ int source_row(int row, int rows, int row_order) {
    if (row_order == 0) return row;
    return row < rows / 2 ? 2 * row : 2 * (row - rows / 2) + 1;
}

// One thread per 8-byte granule of the tiled image.
__global__ __launch_bounds__(256) void repack_q_kernel(const uint2* __restrict__ q, int rows, int cols, int row_order,
                                                       uint2* __restrict__ qt) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)rows * cols / 16;
    if (o >= total) return;
    const int NB = cols / 64, P = rows / 32;
    const int half = (int)(o & 1);
    const int lane = (int)((o >> 1) & 63);
    const int64_t pb = o >> 7;
    const int b = (int)(pb % NB), p = (int)(pb / NB);
    const int r = lane & 15, g = lane >> 4;
    const int row = source_row(16 * (p + half * P) + r, rows, row_order);
    const int64_t src_byte = ((int64_t)row * cols + 64 * b + 16 * g) / 2;
    qt[o] = q[src_byte / 8];
}

__global__ __launch_bounds__(256) void repack_absmax_kernel(const float* __restrict__ a, int rows, int cols, int row_order,
                                                            float* __restrict__ at) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int NB = cols / 64, P = rows / 32;
    const int64_t total = (int64_t)rows * NB;
    if (o >= total) return;
    const int i = (int)(o & 3);
    int64_t t = o >> 2;
    const int b = (int)(t % NB);
    t /= NB;
    const int g = (int)(t & 3);
    t >>= 2;
    const int half = (int)(t & 1);
    const int p = (int)(t >> 1);
    const int row = source_row(16 * (p + half * P) + 4 * g + i, rows, row_order);
    at[o] = a[(int64_t)row * NB + b];
}

__global__ __launch_bounds__(256) void repack_absmax_u8_kernel(const uint8_t* __restrict__ a, int rows, int cols, int row_order,
                                                               uint8_t* __restrict__ at) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int NB = cols / 64, P = rows / 32;
    const int64_t total = (int64_t)rows * NB;
    if (o >= total) return;
    const int i = (int)(o & 3);
    int64_t t = o >> 2;
    const int b = (int)(t % NB);
    t /= NB;
    const int g = (int)(t & 3);
    t >>= 2;
    const int half = (int)(t & 1);
    const int p = (int)(t >> 1);
    const int row = source_row(16 * (p + half * P) + 4 * g + i, rows, row_order);
    at[o] = a[(int64_t)row * NB + b];
}

}  // namespace m355

using namespace m355;

extern "C" int m355_nf4_dequant(const uint8_t* q, const float* absmax, int64_t n, int blocksize, void* out_f16,
                                m355_stream_t stream) {
    if (!q || !absmax || !out_f16) return fail(-1, "m355_nf4_dequant: null pointer");
    if (n <= 0 || blocksize <= 0 || (blocksize % 8) || (n % blocksize))
        return fail(-2, "m355_nf4_dequant: need n %% blocksize == 0 and blocksize %% 8 == 0 (n=%lld, blocksize=%d)",
                    (long long)n, blocksize);
    if (!aligned(q, 4) || !aligned(out_f16, 16)) return fail(-3, "m355_nf4_dequant: q must be 4-byte and out 16-byte aligned");
    const int64_t items = n / 8;
    const int ipb = blocksize / 8;
    int shift = -1;
    if ((ipb & (ipb - 1)) == 0) { shift = 0; while ((1 << shift) < ipb) shift++; }
    // persistent: up to two workgroups per CU (their code tables fill the LDS), each wave walking 1 KiB-in steps
    int dev = 0, cus = 0;
    if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (int rc = hip_status(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
    if (cus <= 0) cus = 256;
    int64_t blocks = (items + DQ_THREADS * 4 - 1) / (DQ_THREADS * 4);
    if (blocks > 2 * cus) blocks = 2 * cus;
    nf4_dequant_kernel<<<dim3((unsigned)blocks), dim3(DQ_THREADS), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const uint32_t*>(q), absmax, items, ipb, shift, reinterpret_cast<u32x4_t*>(out_f16));
    M355_LAUNCH_CHECK("m355_nf4_dequant");
    return 0;
}

extern "C" int m355_repack_nf4(const uint8_t* q, const float* absmax, int rows, int cols, uint8_t* q_tiled,
                               float* absmax_tiled, m355_stream_t stream) {
    return m355_repack_nf4_ex(q, absmax, rows, cols, 0, q_tiled, absmax_tiled, stream);
}

extern "C" int m355_repack_nf4_ex(const uint8_t* q, const float* absmax, int rows, int cols, int row_order,
                                  uint8_t* q_tiled, float* absmax_tiled, m355_stream_t stream) {
    if (row_order != 0 && row_order != 1) return fail(-2, "m355_repack_nf4: row_order %d (0 = as is, 1 = interleaved pairs)", row_order);
    if (!q || !absmax || !q_tiled || !absmax_tiled) return fail(-1, "m355_repack_nf4: null pointer");
    if (rows <= 0 || cols <= 0 || (rows % 32) || (cols % 64))
        return fail(-2, "m355_repack_nf4: need rows %% 32 == 0 and cols %% 64 == 0 (rows=%d, cols=%d)", rows, cols);
    if (!aligned(q, 8) || !aligned(q_tiled, 16)) return fail(-3, "m355_repack_nf4: q 8-byte, q_tiled 16-byte aligned");
    if (q == q_tiled || absmax == absmax_tiled) return fail(-4, "m355_repack_nf4: in-place repack is not supported");
    const int64_t granules = (int64_t)rows * cols / 16;
    repack_q_kernel<<<dim3((unsigned)((granules + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const uint2*>(q), rows, cols, row_order, reinterpret_cast<uint2*>(q_tiled));
    M355_LAUNCH_CHECK("m355_repack_nf4(q)");
    const int64_t na = (int64_t)rows * (cols / 64);
    repack_absmax_kernel<<<dim3((unsigned)((na + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(absmax, rows, cols, row_order,
                                                                                                   absmax_tiled);
    M355_LAUNCH_CHECK("m355_repack_nf4(absmax)");
    return 0;
}

extern "C" int m355_repack_absmax_u8(const uint8_t* absmax_u8, int rows, int cols, int row_order, uint8_t* absmax_tiled,
                                     m355_stream_t stream) {
    if (!absmax_u8 || !absmax_tiled) return fail(-1, "m355_repack_absmax_u8: null pointer");
    if (rows <= 0 || cols <= 0 || (rows % 32) || (cols % 64) || (row_order != 0 && row_order != 1))
        return fail(-2, "m355_repack_absmax_u8: need rows %% 32 == 0, cols %% 64 == 0, row_order 0 or 1 (rows=%d, cols=%d)", rows, cols);
    if (absmax_u8 == absmax_tiled) return fail(-4, "m355_repack_absmax_u8: in-place repack is not supported");
    const int64_t na = (int64_t)rows * (cols / 64);
    repack_absmax_u8_kernel<<<dim3((unsigned)((na + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(absmax_u8, rows, cols,
                                                                                                      row_order, absmax_tiled);
    M355_LAUNCH_CHECK("m355_repack_absmax_u8");
    return 0;
}
