// Proposal directions of the BLOCKED proposer for any d <= 32 (gfx950): parameter blocks
// sorted slow -> fast, per-block oversampling, and the slow / fast sequences of the dragging
// step.  One 64-lane workgroup per (group, cycle); dimension-independent (run-time d), since
// this kernel is a few per cent of a launch.
//
// Restates (paths relative to the reference checkout):
//   cobaya/samplers/mcmc/proposal.py:32-55 (CyclicIndexRandomizer), 58-93 (direction
//   proposers), 176-224 (BlockedProposer cyclers and get_block_proposal), 226-260 (transforms)
//   cobaya/functions.py:35-61 (Haar basis)
// in the ensemble form fixed by oracle/mcmc_oracle.c (orc_block_schedule, orc_basis_blocked):
// the HIP output must equal the oracle's bit for bit.
#include "det_math.h"
#include "kernels.h"

namespace mcmc {
namespace {

constexpr uint32_t kStreamPerm = 2u;
constexpr int kMaxSlots = 2048;   // slots per cycle this kernel supports
constexpr int kMaxN = 32;

__global__ void __launch_bounds__(64) basis_blocked_kernel(const BlockedBasisArgs a)
{
    __shared__ double sT[kMaxN * (kMaxN + 1)];   // leading dimension d | 1 (odd: no conflicts)
    __shared__ short sslot[kMaxN];               // slot of column c of the basis at hand
    __shared__ double sH[kMaxN * (kMaxN + 1)];
    __shared__ double sz[(kMaxN + 2) * (kMaxN - 1) / 2 + 2];
    __shared__ double sx[kMaxN + 1];
    __shared__ double sDn[kMaxN], sPv[kMaxN], sDen[kMaxN];
    // per-slot arrays sized by the cycle length (dynamic LDS): several workgroups per CU
    extern __shared__ unsigned dyn_lds[];
    unsigned* const sw = dyn_lds;                       // [L] shuffle words
    short* const sblk = (short*)(dyn_lds + a.L);        // [L] block of the slot
    short* const sbas = sblk + a.L;                     // [L] basis number
    short* const scol = sbas + a.L;                     // [L] column of the basis
    __shared__ int sIofJ[kMaxN], sSize[kMaxN], sOver[kMaxN];
    const int t = threadIdx.x;
    const int d = a.d, L = a.L;
    const int pg = blockIdx.x / a.ncyc, pc = blockIdx.x % a.ncyc;
    const uint32_t group = a.group0 + (uint32_t)pg;
    const uint32_t cycle = a.cycle0 + (uint32_t)pc;
    double* __restrict__ Vout = a.V + ((size_t)pg * a.ncyc + pc) * a.slab;
    int* __restrict__ Fout = a.vflag ? a.vflag + ((size_t)pg * a.ncyc + pc) * L : nullptr;

    const int ldt = d | 1;
    for (int i = t; i < d * d; i += 64) sT[(i / d) * ldt + i % d] = a.T[i];
    if (t < d) sIofJ[t] = a.i_of_j[t];
    if (t < a.n_blocks) {
        sSize[t] = a.block_size[t];
        sOver[t] = a.oversample[t];
    }
    // Philox words of the shuffle, in parallel; the swaps themselves are sequential
    if (L > 2)
        for (int i = 1 + t; i < L; i += 64)
            sw[i] = philox4x32_10(a.key0, a.key1, group, kStreamPerm | ((uint32_t)a.which << 8),
                                  cycle, (uint32_t)i).w0;
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int b = 0; b < a.n_blocks; ++b) {
            int reps;
            if (a.which == 0) reps = sOver[b] * sSize[b];
            else if (a.which == 1) reps = (b <= a.drag_last_slow) ? sSize[b] : 0;
            else reps = (b > a.drag_last_slow) ? sSize[b] : 0;
            for (int r = 0; r < reps; ++r) sblk[n++] = (short)b;
        }
        if (L > 2)
            for (int i = L - 1; i >= 1; --i) {
                const int j = (int)(((unsigned long long)sw[i] * (unsigned long long)(i + 1)) >> 32);
                const short tmp = sblk[i];
                sblk[i] = sblk[j];
                sblk[j] = tmp;
            }
        int used[kMaxN];
        for (int b = 0; b < a.n_blocks; ++b) used[b] = 0;
        for (int s = 0; s < L; ++s) {
            const int b = sblk[s], n_b = sSize[b];
            sbas[s] = (short)(used[b] / n_b);
            scol[s] = (short)(used[b] % n_b);
            ++used[b];
        }
    }
    __syncthreads();

    int jb = 0;
    for (int b = 0; b < a.n_blocks; jb += sSize[b], ++b) {
        const int n = sSize[b];
        int reps;
        if (a.which == 0) reps = sOver[b] * n;
        else if (a.which == 1) reps = (b <= a.drag_last_slow) ? n : 0;
        else reps = (b > a.drag_last_slow) ? n : 0;
        if (reps == 0) continue;
        if (n == 1) {  // RandProposer1D: the direction is the block's column of T itself
            for (int s = 0; s < L; ++s)
                if (sblk[s] == b) {
                    if (t < d) Vout[(size_t)s * d + sIofJ[t]] = (t >= jb) ? sT[t * ldt + jb] : 0.0;
                    if (Fout && t == 0) Fout[s] = 1;
                }
            continue;
        }
        const int nz = (n + 2) * (n - 1) / 2;
        const int ldh = n | 1;
        for (int q = 0; q < reps / n; ++q) {
            __syncthreads();
            // Box-Muller normals on the basis stream of (block, basis number)
            for (int j = t; 2 * j < nz; j += 64) {
                const u32x4 w4 = philox4x32_10(
                    a.key0, a.key1, group,
                    kStreamBasis | ((uint32_t)a.which << 4) | ((uint32_t)b << 8), cycle,
                    ((uint32_t)q << 16) | (uint32_t)j);
                const uint64_t ka = ((uint64_t)w4.w0 << 20) | (w4.w1 >> 12);
                const uint64_t kb = ((uint64_t)w4.w2 << 20) | (w4.w3 >> 12);
                const double rad = sqrt(-2.0 * dlog(u52(ka)));
                double sn, cs;
                sincos2pi(kb, sn, cs);
                sz[2 * j] = rad * cs;
                sz[2 * j + 1] = rad * sn;
            }
            __syncthreads();
            // Householder construction, the arithmetic and order of orc_haar_from_normals.
            // The scalars of reflection m0 (norm, sign, pivot, denominator) depend on the
            // normals only: thread m0 forms them, all reflections in parallel; row t of H
            // lives in registers (kMaxN entries, zero beyond n, the reflector zero-padded so
            // that the extra terms are exact no-ops) and goes to LDS for the column products.
            if (t < n - 1) {
                const int m = n - t;
                const int ix0 = t * n - (t * (t - 1)) / 2;
                double norm2 = 0.0;
                for (int k = 0; k < m; ++k) norm2 = fma(sz[ix0 + k], sz[ix0 + k], norm2);
                const double x0 = sz[ix0];
                const double Dn = (x0 < 0.0) ? -1.0 : 1.0;
                const double x0n = x0 + Dn * sqrt(norm2);
                double tt = norm2 - x0 * x0;
                tt = tt + x0n * x0n;
                sDn[t] = Dn;
                sPv[t] = x0n;
                sDen[t] = sqrt(0.5 * tt);
            }
            __syncthreads();
            double Dmine = (t < n - 1) ? sDn[t] : 1.0;
            double dprod = 1.0;
            for (int m0 = 0; m0 < n - 1; ++m0) dprod *= sDn[m0];
            if (t == n - 1) Dmine = (((n - 1) & 1) ? -1.0 : 1.0) * dprod;
            double h[kMaxN];
#pragma unroll
            for (int k = 0; k < kMaxN; ++k) h[k] = (k == t) ? 1.0 : 0.0;
            int ix = 0;
#pragma unroll
            for (int m0 = 0; m0 < kMaxN - 1; ++m0) {
                if (m0 < n - 1) {   // uniform
                    const int m = n - m0;
                    const double x0n = sPv[m0], den = sDen[m0];
                    __syncthreads();
                    if (t < kMaxN - m0) sx[t] = (t < m) ? ((t == 0) ? x0n : sz[ix + t]) / den : 0.0;
                    __syncthreads();
                    double tmp = 0.0;
#pragma unroll
                    for (int k = 0; k < kMaxN - m0; ++k) tmp = fma(h[m0 + k], sx[k], tmp);
#pragma unroll
                    for (int k = 0; k < kMaxN - m0; ++k) h[m0 + k] = fma(-tmp, sx[k], h[m0 + k]);
                    ix += m;
                }
            }
            if (t < n) {
#pragma unroll
                for (int k = 0; k < kMaxN; ++k)
                    if (k < n) sH[t * ldh + k] = Dmine * h[k];
            }
            __syncthreads();
            // The columns of this basis, wherever the shuffle put them: all n * d outputs
            // (column c, sorted row j) are spread over the 64 lanes, four independent chains
            // per lane in flight.  Every chain runs over the whole block, k = 0 .. n-1: T is
            // lower triangular, so the terms beyond the oracle's k <= j - j_b (and the rows
            // above the block) multiply exact zeros and leave the sum -- or +0.0 -- unchanged.
            for (int s = t; s < L; s += 64)
                if (sblk[s] == b && sbas[s] == q) {
                    sslot[scol[s]] = (short)s;
                    if (Fout) Fout[s] = 0;
                }
            __syncthreads();
            const int n_out = n * d;
            for (int o0 = t; o0 < n_out; o0 += 256) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                int cc[4], jj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o = o0 + 64 * u < n_out ? o0 + 64 * u : o0;
                    cc[u] = o / d;
                    jj[u] = o % d;
                }
                for (int k = 0; k < n; ++k) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[u] = fma(sT[jj[u] * ldt + jb + k], sH[k * ldh + cc[u]], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (o0 + 64 * u < n_out)
                        Vout[(size_t)sslot[cc[u]] * d + sIofJ[jj[u]]] = acc[u];
            }
        }
    }
}


// ---------------------------------------------------------------- the same for 32 < d <= 128
// Run-time d and block sizes up to 128, one 256-thread workgroup per (group, cycle).  Written
// for coverage, not speed (the directions of a blocked cycle are a few per cent of a launch):
// H of the block at hand sits in LDS, thread i owns row i; the normals of a reflection are
// drawn when the reflection needs them (each element from its Box-Muller pair, so nothing of
// size n^2 / 2 is kept); the per-reflection scalars are one sequential chain on thread 0.  The
// arithmetic and its order are orc_basis_blocked / orc_haar_from_normals: a block of n <= 32
// parameters projects with ONE chain, a larger one with four interleaved chains.
__global__ void __launch_bounds__(256) basis_blocked_big_kernel(const BlockedBasisArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_raw[];
    const int t = threadIdx.x;
    const int d = a.d, L = a.L, nmax = a.nmax;
    const int ldh = nmax | 1;
    double* const sH = (double*)dyn_raw;                    // [nmax][ldh]
    double* const sx = sH + (size_t)nmax * ldh;             // [nmax + 1] reflector
    double* const sD = sx + nmax + 1;                       // [nmax] signs D_n
    unsigned* const sw = (unsigned*)(sD + nmax);            // [L] shuffle words
    short* const sblk = (short*)(sw + L);                   // [L] block of the slot
    short* const sbas = sblk + L;                           // [L] basis number
    short* const scol = sbas + L;                           // [L] column of the basis
    __shared__ int sIofJ[128], sSize[32], sOver[32];
    __shared__ double sscal[2];                             // pivot, denominator
    const int pg = blockIdx.x / a.ncyc, pc = blockIdx.x % a.ncyc;
    const uint32_t group = a.group0 + (uint32_t)pg;
    const uint32_t cycle = a.cycle0 + (uint32_t)pc;
    double* __restrict__ Vout = a.V + ((size_t)pg * a.ncyc + pc) * a.slab;
    int* __restrict__ Fout = a.vflag ? a.vflag + ((size_t)pg * a.ncyc + pc) * L : nullptr;
    const int ld = a.ld;

    if (t < d) sIofJ[t] = a.i_of_j[t];
    if (t < a.n_blocks) {
        sSize[t] = a.block_size[t];
        sOver[t] = a.oversample[t];
    }
    if (L > 2)
        for (int i = 1 + t; i < L; i += 256)
            sw[i] = philox4x32_10(a.key0, a.key1, group, kStreamPerm | ((uint32_t)a.which << 8),
                                  cycle, (uint32_t)i).w0;
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int b = 0; b < a.n_blocks; ++b) {
            int reps;
            if (a.which == 0) reps = sOver[b] * sSize[b];
            else if (a.which == 1) reps = (b <= a.drag_last_slow) ? sSize[b] : 0;
            else reps = (b > a.drag_last_slow) ? sSize[b] : 0;
            for (int r = 0; r < reps; ++r) sblk[n++] = (short)b;
        }
        if (L > 2)
            for (int i = L - 1; i >= 1; --i) {
                const int j = (int)(((unsigned long long)sw[i] * (unsigned long long)(i + 1)) >> 32);
                const short tmp = sblk[i];
                sblk[i] = sblk[j];
                sblk[j] = tmp;
            }
        int used[32];
        for (int b = 0; b < a.n_blocks; ++b) used[b] = 0;
        for (int s = 0; s < L; ++s) {
            const int b = sblk[s], n_b = sSize[b];
            sbas[s] = (short)(used[b] / n_b);
            scol[s] = (short)(used[b] % n_b);
            ++used[b];
        }
    }
    __syncthreads();

    int jb = 0;
    for (int b = 0; b < a.n_blocks; jb += sSize[b], ++b) {
        const int n = sSize[b];
        int reps;
        if (a.which == 0) reps = sOver[b] * n;
        else if (a.which == 1) reps = (b <= a.drag_last_slow) ? n : 0;
        else reps = (b > a.drag_last_slow) ? n : 0;
        if (reps == 0) continue;
        if (n == 1) {  // RandProposer1D: the direction is the block's column of T itself
            for (int s = 0; s < L; ++s)
                if (sblk[s] == b) {
                    if (t < d) Vout[(size_t)s * ld + sIofJ[t]] = (t >= jb) ? a.T[t * d + jb] : 0.0;
                    if (Fout && t == 0) Fout[s] = 1;
                }
            continue;
        }
        const bool four = n > 32;   // projection order of orc_haar_from_normals
        for (int q = 0; q < reps / n; ++q) {
            __syncthreads();
            for (int e = t; e < n * ldh; e += 256) sH[e] = 0.0;
            __syncthreads();
            if (t < n) sH[t * ldh + t] = 1.0;
            int ix = 0;
            double dprod = 1.0;
            for (int m0 = 0; m0 < n - 1; ++m0) {
                const int m = n - m0;
                __syncthreads();
                // the m normals of this reflection: element e = ix + k of the basis stream
                for (int k = t; k < m; k += 256) {
                    const int e = ix + k;
                    const u32x4 w4 = philox4x32_10(
                        a.key0, a.key1, group,
                        kStreamBasis | ((uint32_t)a.which << 4) | ((uint32_t)b << 8), cycle,
                        ((uint32_t)q << 16) | (uint32_t)(e >> 1));
                    const uint64_t ka = ((uint64_t)w4.w0 << 20) | (w4.w1 >> 12);
                    const uint64_t kb = ((uint64_t)w4.w2 << 20) | (w4.w3 >> 12);
                    const double rad = sqrt(-2.0 * dlog(u52(ka)));
                    double sn, cs;
                    sincos2pi(kb, sn, cs);
                    sx[k] = (e & 1) ? rad * sn : rad * cs;
                }
                __syncthreads();
                if (t == 0) {
                    double norm2 = 0.0;
                    for (int k = 0; k < m; ++k) norm2 = fma(sx[k], sx[k], norm2);
                    const double x0 = sx[0];
                    const double Dn = (x0 < 0.0) ? -1.0 : 1.0;
                    const double x0n = x0 + Dn * sqrt(norm2);
                    double tt = norm2 - x0 * x0;
                    tt = tt + x0n * x0n;
                    sD[m0] = Dn;
                    sscal[0] = x0n;
                    sscal[1] = sqrt(0.5 * tt);
                }
                __syncthreads();
                dprod *= sD[m0];
                {
                    const double den = sscal[1];
                    for (int k = t; k < m; k += 256) sx[k] = ((k == 0) ? sscal[0] : sx[k]) / den;
                }
                __syncthreads();
                if (t < n) {
                    double* __restrict__ row = sH + t * ldh + m0;
                    double tc[4] = {0.0, 0.0, 0.0, 0.0};
                    if (four)
                        for (int k = 0; k < m; ++k) tc[(m0 + k) & 3] = fma(row[k], sx[k], tc[(m0 + k) & 3]);
                    else
                        for (int k = 0; k < m; ++k) tc[0] = fma(row[k], sx[k], tc[0]);
                    const double tmp = four ? (tc[0] + tc[1]) + (tc[2] + tc[3]) : tc[0];
                    for (int k = 0; k < m; ++k) row[k] = fma(-tmp, sx[k], row[k]);
                }
                ix += m;
            }
            __syncthreads();
            if (t < n) {
                const double Dmine = (t < n - 1) ? sD[t] : (((n - 1) & 1) ? -1.0 : 1.0) * dprod;
                for (int k = 0; k < n; ++k) sH[t * ldh + k] = Dmine * sH[t * ldh + k];
            }
            __syncthreads();
            // the columns of this basis, wherever the shuffle put them: item = (slot, sorted row)
            for (int s = 0; s < L; ++s) {
                if (sblk[s] != b || sbas[s] != q) continue;   // uniform
                const int c = scol[s];
                for (int j = t; j < d; j += 256) {   // (rows above the block: zero)
                    const int kmax = j - jb < n - 1 ? j - jb : n - 1;
                    double acc = 0.0;
                    for (int k = 0; k <= kmax; ++k)
                        acc = fma(a.T[(size_t)j * d + jb + k], sH[k * ldh + c], acc);
                    Vout[(size_t)s * ld + sIofJ[j]] = acc;
                }
                if (Fout && t == 0) Fout[s] = 0;
            }
        }
    }
}

}  // namespace
}  // namespace mcmc

extern "C" hipError_t mcmc_hip_launch_blocked_basis(const mcmc::BlockedBasisArgs* a, int n_groups,
                                                    hipStream_t st)
{
    if (a->L > mcmc::kMaxSlots || a->n_blocks > mcmc::kMaxN) return hipErrorInvalidValue;
    if (a->d > mcmc::kMaxN) {   // 32 < d <= 128: the general kernel
        if (a->d > 128 || a->nmax < 1 || a->nmax > 128) return hipErrorInvalidValue;
        const size_t ldh = (size_t)(a->nmax | 1);
        const size_t lds = sizeof(double) * ((size_t)a->nmax * ldh + 2 * (size_t)a->nmax + 1) +
                           (size_t)a->L * (sizeof(unsigned) + 3 * sizeof(short)) + 32;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)mcmc::basis_blocked_big_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(mcmc::basis_blocked_big_kernel, dim3(n_groups * a->ncyc), dim3(256), lds,
                           st, *a);
        return hipGetLastError();
    }
    const size_t lds = (size_t)a->L * (sizeof(unsigned) + 3 * sizeof(short)) + 16;
    hipLaunchKernelGGL(mcmc::basis_blocked_kernel, dim3(n_groups * a->ncyc), dim3(64), lds, st, *a);
    return hipGetLastError();
}
