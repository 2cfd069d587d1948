// csrc/kernels_witness.hip -- R1CS witness generation for the reference's AES circuit and the sparse matrix products.
//
// Replaces the value side of the gadget synthesis in /root/reference/src/lib.rs:176-293 and src/aes_circuit.rs:20-427
// (every UInt8/Boolean op there both allocates a variable and computes its value on the CPU): here
//   k_aes_trace       one lane per ECB block: plain AES-128 with every intermediate byte the circuit names written to the
//                     per-proof trace buffer (layout: trace_layout.h),
//   k_witness_expand  one lane per column of z: decode the variable's descriptor (compiled once by circuit.cpp) and gather
//                     its bit -- S-box mux-tree variables are a table lookup S[(node << (level+1)) | (x & mask)],
//   k_spmv_bits       z_A = A z, z_B = B z over 0/1 assignments with small integer coefficients (ark-marlin prover_init),
//   k_t_evals         the round-2 "t" accumulation through a column-bucketed copy of A, B, C.
#include <mutex>
#include "hip_util.hpp"
#include "trace_layout.h"

namespace zk {
namespace gpu {

#define GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// device copies of the S-box table, one per HIP device (keys on several GPUs of one process each use their own)
static uint8_t *g_sbox_dev[64] = {nullptr};
static std::mutex g_sbox_mu;
static uint8_t *sbox_here() {
    int d = 0;
    HIP_CHECK(hipGetDevice(&d));
    if (d < 0 || d >= 64) throw GpuError("device ordinal out of range");
    std::lock_guard<std::mutex> g(g_sbox_mu);
    return g_sbox_dev[d];
}
void upload_sbox(const uint8_t table[256]) {
    int d = 0;
    HIP_CHECK(hipGetDevice(&d));
    if (d < 0 || d >= 64) throw GpuError("device ordinal out of range");
    std::lock_guard<std::mutex> g(g_sbox_mu);
    if (!g_sbox_dev[d]) g_sbox_dev[d] = (uint8_t *)dmalloc(256);
    HIP_CHECK(hipMemcpy(g_sbox_dev[d], table, 256, hipMemcpyHostToDevice));
}

__device__ __forceinline__ uint8_t xtime(uint8_t c) { return (uint8_t)((c << 1) ^ (((c >> 7) & 1) * 0x1B)); }

// grid: nproofs * (nblocks + 1) lanes; lane (p, 0) writes the key schedule part, lane (p, 1 + b) block b
__global__ void k_aes_trace(uint8_t *__restrict__ trace, size_t stride, const uint8_t *__restrict__ msgs, const uint8_t *__restrict__ keys, uint32_t nproofs,
                            uint32_t nblocks, const uint8_t *__restrict__ sbox) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nproofs * (nblocks + 1)) return;
    uint32_t p = t / (nblocks + 1), which = t % (nblocks + 1);
    uint8_t *tr = trace + (size_t)p * stride;
    const uint8_t *key = keys + 16 * (size_t)p;
    // key schedule (src/aes_circuit.rs:83-113): words big-endian, RotWord = bytes rotate-left 1
    uint8_t w[44][4];
    const uint8_t rc[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};
    for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) w[i][k] = key[4 * i + k];
    for (int i = 4; i < 44; i++) {
        if (i % 4 == 0) {
            int q = i / 4 - 1;
            uint8_t sub[4], pre[4];
            for (int k = 0; k < 4; k++) sub[k] = sbox[w[i - 1][(k + 1) & 3]];
            for (int k = 0; k < 4; k++) { pre[k] = w[i - 4][k] ^ sub[k]; w[i][k] = pre[k]; }
            w[i][0] ^= rc[q];
            if (which == 0) for (int k = 0; k < 4; k++) { tr[TR_KS_SUB + 4 * q + k] = sub[k]; tr[TR_KS_PRE + 4 * q + k] = pre[k]; }
        } else {
            for (int k = 0; k < 4; k++) w[i][k] = w[i - 4][k] ^ w[i - 1][k];
        }
    }
    if (which == 0) {
        for (int i = 0; i < 16; i++) tr[TR_KEY + i] = key[i];
        for (int i = 0; i < 44; i++) for (int k = 0; k < 4; k++) tr[TR_KS_W + 4 * i + k] = w[i][k];
        return;
    }
    uint32_t b = which - 1;
    uint8_t *bl = tr + TR_BLOCK0 + (size_t)b * TR_BLOCK_STRIDE;
    const uint8_t *msg = msgs + ((size_t)p * nblocks + b) * 16;
    uint8_t s[16], u[16], v[16];
    for (int i = 0; i < 16; i++) { bl[TR_BL_MSG + i] = msg[i]; s[i] = msg[i] ^ key[i]; bl[TR_BL_S + i] = s[i]; }
    for (int r = 1; r <= 10; r++) {
        for (int i = 0; i < 16; i++) { v[i] = sbox[s[i]]; bl[TR_BL_SB + 16 * (r - 1) + i] = v[i]; }
        for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) u[4 * c + rr] = v[4 * ((c + rr) & 3) + rr];     // ShiftRows
        if (r <= 9) {
            for (int c = 0; c < 4; c++) {
                uint8_t a[4], xb[4];
                for (int k = 0; k < 4; k++) { a[k] = u[4 * c + k]; xb[k] = xtime(a[k]); bl[TR_BL_XT + 16 * (r - 1) + 4 * c + k] = xb[k]; }
                // left-assoc xor chains of src/aes_circuit.rs:391-426
                const uint8_t term[4][5] = {{xb[0], a[3], a[2], xb[1], a[1]}, {xb[1], a[0], a[3], xb[2], a[2]}, {xb[2], a[1], a[0], xb[3], a[3]}, {xb[3], a[2], a[1], xb[0], a[0]}};
                for (int o = 0; o < 4; o++) {
                    uint8_t acc = term[o][0];
                    for (int q = 1; q < 5; q++) { acc ^= term[o][q]; bl[TR_BL_MP + 64 * (r - 1) + 4 * (4 * c + o) + (q - 1)] = acc; }
                    v[4 * c + o] = acc;
                }
            }
        } else {
            for (int i = 0; i < 16; i++) v[i] = u[i];
        }
        for (int i = 0; i < 16; i++) { s[i] = v[i] ^ w[4 * r + i / 4][i % 4]; bl[TR_BL_S + 16 * r + i] = s[i]; }
    }
}
void aes_trace(uint8_t *trace, size_t stride, const uint8_t *msgs, const uint8_t *keys, uint32_t nproofs, uint32_t nblocks, stream_t s) {
    uint8_t *g_sbox = sbox_here();
    if (!g_sbox) throw GpuError("aes_trace: S-box table not uploaded on this device");
    uint32_t lanes = nproofs * (nblocks + 1);
    hipLaunchKernelGGL(k_aes_trace, dim3((lanes + 63) / 64), dim3(64), 0, (hipStream_t)s, trace, stride, msgs, keys, nproofs, nblocks, g_sbox);
    HIP_LAUNCH_CHECK();
}

__global__ void k_witness_expand(uint8_t *__restrict__ z, const uint32_t *__restrict__ desc, uint32_t ncols, const uint8_t *__restrict__ trace,
                                 const uint32_t *__restrict__ sbox_in_off, const uint32_t *__restrict__ sbox_tmpl, const uint8_t *__restrict__ sbox) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    uint32_t d = desc[i], kind = d >> WD_KIND_SHIFT, bit;
    if (kind == WD_BYTEBIT) {
        uint32_t off = (d >> 4) & 0x3ffffff, b = (d >> 1) & 7;
        bit = ((trace[off] >> b) & 1) ^ (d & 1);
    } else if (kind == WD_SBOX) {
        uint32_t inst = (d >> 11) & 0x7ffff, te = sbox_tmpl[(d >> 1) & 0x3ff];
        uint32_t lvl = (te >> 12) & 7, node = (te >> 4) & 0xff, b = (te >> 1) & 7;
        uint32_t x = trace[sbox_in_off[inst]];
        uint32_t idx = (node << (lvl + 1)) | (x & ((2u << lvl) - 1));
        bit = ((sbox[idx] >> b) & 1) ^ (d & 1);
    } else {
        bit = d & 1;
    }
    z[i] = (uint8_t)bit;
}
void witness_expand(uint8_t *z, const uint32_t *desc, uint32_t ncols, const uint8_t *trace, const uint32_t *sbox_in_off, const uint32_t *sbox_tmpl, stream_t s) {
    uint8_t *g_sbox = sbox_here();
    if (!g_sbox) throw GpuError("witness_expand: S-box table not uploaded on this device");
    hipLaunchKernelGGL(k_witness_expand, GRID(ncols), 0, (hipStream_t)s, z, desc, ncols, trace, sbox_in_off, sbox_tmpl, g_sbox);
    HIP_LAUNCH_CHECK();
}

__device__ __forceinline__ F small_to_field(long long v) { return F::from_i64(v); }

__global__ void k_spmv_bits(F *__restrict__ out, int8_t *__restrict__ small_out, size_t rows_out, const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col, const int64_t *__restrict__ coeff,
                            size_t rows, const uint8_t *__restrict__ z) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_out) return;
    long long acc = 0;
    if (r < rows) for (uint32_t i = rowptr[r]; i < rowptr[r + 1]; i++) acc += z[col[i]] ? coeff[i] : 0;
    out[r] = acc == 0 ? F::zero() : small_to_field(acc);
    if (small_out) small_out[r] = (int8_t)(acc > 127 ? 127 : (acc < -127 ? -127 : acc));
}
void spmv_bits(F *out, int8_t *small_out, size_t rows_out, const uint32_t *rowptr, const uint32_t *col, const int64_t *coeff, size_t rows, const uint8_t *z, stream_t s) {
    hipLaunchKernelGGL(k_spmv_bits, GRID(rows_out), 0, (hipStream_t)s, out, small_out, rows_out, rowptr, col, coeff, rows, z); HIP_LAUNCH_CHECK();
}
__global__ void k_w_classes(int8_t *__restrict__ out, const uint8_t *__restrict__ z, uint32_t n, uint32_t m, uint32_t num_witness) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t ratio = n / m;
    if (k % ratio == 0) { out[k] = 0; return; }
    uint32_t wi = k - k / ratio - 1;
    out[k] = (wi < num_witness && z[m + wi]) ? 1 : 0;
}
void w_classes(int8_t *out, const uint8_t *z, uint32_t n, uint32_t m, uint32_t num_witness, stream_t s) {
    hipLaunchKernelGGL(k_w_classes, GRID(n), 0, (hipStream_t)s, out, z, n, m, num_witness); HIP_LAUNCH_CHECK();
}
__global__ void k_lagrange_finish(F *__restrict__ lag, F *__restrict__ lag_w, const F *__restrict__ elems, uint32_t n, uint32_t m, F vx_inv, F n_inv) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    F v = lag[k] * (elems[k] * n_inv);  // lag[k] arrives as (beta^n - 1) / (beta - g^k)
    lag[k] = v;
    lag_w[k] = (k % (n / m) == 0) ? F::zero() : v * vx_inv;
}
void lagrange_scalars(F *lag, F *lag_w, const F *elems, const F &beta, uint32_t n, uint32_t m, stream_t s) {
    int lg_n = 0;
    while ((1u << lg_n) < n) lg_n++;
    F vx_inv = (beta.pow_u64(m) - F::one()).inverse();
    // (beta^n - 1) / (beta - g^k) = r(beta, g^k): the product tree of kernels_poly.hip (no batch inversion)
    const size_t need = vanishing_quotient_scratch(lg_n, 1);
    DevPtr<F> scratch(need);
    F *outs[1] = {lag};
    const F one = F::one();
    vanishing_quotient_evals(outs, &one, 1, beta, elems, n, lg_n, scratch, need, s);
    hipLaunchKernelGGL(k_lagrange_finish, GRID(n), 0, (hipStream_t)s, lag, lag_w, elems, n, m, vx_inv, F::from_u64(n).inverse()); HIP_LAUNCH_CHECK();
    sync(s);                            // (the scratch is released on return)
}

__global__ void k_w_evals(F *__restrict__ out, const uint8_t *__restrict__ z, const F *__restrict__ x_evals, uint32_t n, uint32_t m, uint32_t num_witness) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t ratio = n / m;
    if (k % ratio == 0) { out[k] = F::zero(); return; }
    uint32_t wi = k - k / ratio - 1;
    F w = (wi < num_witness && z[m + wi]) ? F::one() : F::zero();
    out[k] = w - x_evals[k];
}
void w_evals(F *out, const uint8_t *z, const F *x_evals, uint32_t n, uint32_t m, uint32_t num_witness, stream_t s) {
    hipLaunchKernelGGL(k_w_evals, GRID(n), 0, (hipStream_t)s, out, z, x_evals, n, m, num_witness); HIP_LAUNCH_CHECK();
}
__global__ void k_bits_to_field(F *__restrict__ out, const uint8_t *__restrict__ z, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = z[i] ? F::one() : F::zero(); }
void bits_to_field(F *out, const uint8_t *z, size_t n, stream_t s) { if (!n) return; hipLaunchKernelGGL(k_bits_to_field, GRID(n), 0, (hipStream_t)s, out, z, n); HIP_LAUNCH_CHECK(); }

// Two passes so that heavy columns (the constant One, key bits) do not serialize on one lane: pass 1 = one lane per SEGMENT of at most
// T_SEG entries of one column, pass 2 = one lane per column adding its segments' partial sums.
__global__ void k_t_partials(F *__restrict__ partial, uint32_t nseg, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end, const uint32_t *__restrict__ row,
                             const uint8_t *__restrict__ mat, const int64_t *__restrict__ coeff, const F *__restrict__ r_alpha, F eta_a, F eta_b, F eta_c) {
    uint32_t sgi = blockIdx.x * blockDim.x + threadIdx.x;
    if (sgi >= nseg) return;
    F acc = F::zero();
    for (uint32_t i = seg_start[sgi]; i < seg_end[sgi]; i++) {
        F eta = mat[i] == 0 ? eta_a : (mat[i] == 1 ? eta_b : eta_c);
        long long c = coeff[i];
        F term = eta * r_alpha[row[i]];
        if (c == 1) acc = acc + term;
        else if (c == -1) acc = acc - term;
        else acc = acc + term * small_to_field(c);
    }
    partial[sgi] = acc;
}
// columns with more than T_HEAVY_SEGMENTS segments (the constant One, key bits) get a whole workgroup: k_t_heavy
__global__ void k_t_columns(F *__restrict__ out, uint32_t n, const uint32_t *__restrict__ col_seg_ptr, const F *__restrict__ partial) {
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    uint32_t a = col_seg_ptr[h], b = col_seg_ptr[h + 1];
    if (b - a > T_HEAVY_SEGMENTS) return;        // summed by k_t_heavy
    F acc = F::zero();
    for (uint32_t i = a; i < b; i++) acc = acc + partial[i];
    out[h] = acc;
}
__global__ void __launch_bounds__(256) k_t_heavy(F *__restrict__ out, const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ col_seg_ptr, const F *__restrict__ partial) {
    __shared__ F sh[256];
    uint32_t h = heavy[blockIdx.x], t = threadIdx.x;
    F acc = F::zero();
    for (uint32_t i = col_seg_ptr[h] + t; i < col_seg_ptr[h + 1]; i += 256) acc = acc + partial[i];
    sh[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)t < s) sh[t] = sh[t] + sh[t + s]; __syncthreads(); }
    if (t == 0) out[h] = sh[0];
}
void t_evals(F *out, uint32_t n, F *partial, uint32_t nseg, const uint32_t *col_seg_ptr, const uint32_t *seg_start, const uint32_t *seg_end, const uint32_t *heavy, uint32_t n_heavy,
             const uint32_t *row, const uint8_t *mat, const int64_t *coeff, const F *r_alpha, const F &eta_a, const F &eta_b, const F &eta_c, stream_t s) {
    if (nseg) { hipLaunchKernelGGL(k_t_partials, GRID(nseg), 0, (hipStream_t)s, partial, nseg, seg_start, seg_end, row, mat, coeff, r_alpha, eta_a, eta_b, eta_c); HIP_LAUNCH_CHECK(); }
    hipLaunchKernelGGL(k_t_columns, GRID(n), 0, (hipStream_t)s, out, n, col_seg_ptr, partial); HIP_LAUNCH_CHECK();
    if (n_heavy) { hipLaunchKernelGGL(k_t_heavy, dim3(n_heavy), dim3(256), 0, (hipStream_t)s, out, heavy, col_seg_ptr, partial); HIP_LAUNCH_CHECK(); }
}

}  // namespace gpu
}  // namespace zk
