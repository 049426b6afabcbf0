// Micro-benchmark (time-boxed): Levenshtein bit-sliced ACROSS pairs -- each lane holds 32 pairs, one bit per
// pair, a wave 2 048 -- against lev_myers32_snap's one pair per lane (12.7 SIMD cycles per pair on cfg2's lengths).
//
// The cell automaton (Myers 1999, the four delta bits of a DP cell, one bit per pair; strsim.rs:141-160 semantics): with the
// vertical delta arriving from the cell to the left (Pv, Mv), the horizontal one from the cell above (Ph, Mh) and Eq = a_i == b_j,
//     Xv = Eq | Mv;  Xh = Eq | Mh;  Ph' = Mv | ~(Xh | Pv);  Mh' = Pv & Xh;  Pv' = Mh | ~(Xv | Ph);  Mv' = Ph & Xv
// -- six three-input ops -- and Eq = AND over the five character planes of ~(A_k[i] ^ B_k[j]): five more.  All full-rate
// (v_bitop3 / v_or), no bit fills.  What it needs that the lane-per-pair form does not: both strings as planes ACROSS pairs
// (word [position][plane], bit = pair) -- a 32 x 32 x 8-bit transpose per 32 pairs and string -- and 2 048 pairs of like lengths
// per wave (a cell is paid for by every pair of the wave: max la x max lb).
//
// This program measures, per pair and in SIMD cycles at 4 waves per SIMD: (1) the automaton alone for la = lb = L (planes already
// in registers / LDS: row tiles of 8 pattern rows in registers, the tile boundary's horizontal deltas through LDS), (2) the
// transposes alone (32 windows of 32 bytes per lane -> planes across pairs), and checks the automaton against the plain DP.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipolars-strsim_amd/csrc bench_support/micro/core_sliced.hip -o bench_support/micro/core_sliced && bench_support/micro/core_sliced
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "strsim_lane_core.h"

using namespace strsim;
constexpr int NP = 5, TILE = 8;

// planes in global memory: word [(pos * NP + k) * 64 + lane]
template <int L>
__global__ __launch_bounds__(64) void k_sliced(const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb, uint32_t *__restrict__ dist_bits,
                                               unsigned long long *clk, int iters)
{
    const uint32_t lane = threadIdx.x;
    // (the text planes come from global memory, five coalesced words per column: in LDS they would be 40 KB per wave at L = 32)
    __shared__ uint32_t s_h[2][L][64];         // horizontal deltas at the bottom of a row tile, per column (P, M): 16 KB per wave at L = 32
    // the distance per pair as a bit-sliced counter (6 planes): D = L + sum over the last column's rows of (Pv - Mv)
    uint32_t cnt[7];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t fold = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 7; ++q) cnt[q] = 0u;
#pragma unroll 1
        for (int j0 = 0; j0 < L; j0 += TILE) {
            uint32_t B[TILE][NP], Pv[TILE], Mv[TILE];
#pragma unroll
            for (int r = 0; r < TILE; ++r) {
#pragma unroll
                for (int k = 0; k < NP; ++k) B[r][k] = pb[((j0 + r) * NP + k) * 64 + lane] ^ fold;
                Pv[r] = 0xFFFFFFFFu; Mv[r] = 0u; // D(0, j) - D(0, j - 1) = +1
            }
#pragma unroll 1
            for (int i = 0; i < L; ++i) {
                uint32_t A[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) A[k] = pa[(i * NP + k) * 64 + lane] ^ fold;
                // the horizontal delta entering the tile from above: row 0's is +1 (D(i, 0) = i), else the tile above's bottom row
                uint32_t Ph = j0 == 0 ? 0xFFFFFFFFu : s_h[0][i][lane], Mh = j0 == 0 ? 0u : s_h[1][i][lane];
#pragma unroll
                for (int r = 0; r < TILE; ++r) {
                    uint32_t Eq = ~(A[0] ^ B[r][0]);
#pragma unroll
                    for (int k = 1; k < NP; ++k) Eq = bitop3<0x90>(Eq, A[k], B[r][k]); // Eq & ~(A ^ B)
                    const uint32_t Xv = Eq | Mv[r], Xh = Eq | Mh;
                    const uint32_t Pho = bitop3<0xF1>(Mv[r], Xh, Pv[r]); // Mv | ~(Xh | Pv)
                    const uint32_t Mho = Pv[r] & Xh;
                    const uint32_t Pvo = bitop3<0xF1>(Mh, Xv, Ph);       // Mh | ~(Xv | Ph)
                    const uint32_t Mvo = Ph & Xv;
                    Pv[r] = Pvo; Mv[r] = Mvo; Ph = Pho; Mh = Mho;
                }
                s_h[0][i][lane] = Ph; s_h[1][i][lane] = Mh;
            }
            // the last column's vertical deltas of this tile into the bit-sliced counter: +Pv, then -Mv (two's complement, 7 planes)
#pragma unroll
            for (int r = 0; r < TILE; ++r) {
                uint32_t c = Pv[r];
#pragma unroll
                for (int q = 0; q < 7; ++q) { const uint32_t s = cnt[q] ^ c; c &= cnt[q]; cnt[q] = s; }
                uint32_t b = Mv[r];
#pragma unroll
                for (int q = 0; q < 7; ++q) { const uint32_t s = cnt[q] ^ b; b &= ~cnt[q]; cnt[q] = s; }
            }
        }
        fold = (cnt[0] & cnt[1] & cnt[2] & cnt[3] & cnt[4] & cnt[5] & cnt[6]) & (it == 0x7FFFFFFF ? 1u : 0u); // (a dependency, always 0)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int q = 0; q < 7; ++q) dist_bits[(blockIdx.x * 7 + q) * 64 + lane] = cnt[q];
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

// 32 windows of 32 bytes per lane (pair p's window in w[p][0..7]) -> planes across pairs: out[pos][k] bit p = bit k of byte pos of pair p.
// Per pair the lane-per-pair plane build (build_planes<5>: bit i of P_k = bit k of byte i), then a 32 x 32 bit transpose per plane.
__device__ __forceinline__ void transpose32(uint32_t (&m)[32])
{
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const uint32_t mask = s == 16 ? 0x0000FFFFu : (s == 8 ? 0x00FF00FFu : (s == 4 ? 0x0F0F0F0Fu : (s == 2 ? 0x33333333u : 0x55555555u)));
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if ((i & s) == 0) {
                const uint32_t a = m[i], b = m[i + s];
                m[i] = bitop3<0xCA>(mask, a, b << s);     // mask ? a : b << s
                m[i + s] = bitop3<0xCA>(mask, a >> s, b); // mask ? a >> s : b
            }
        }
    }
}
__global__ __launch_bounds__(64) void k_transpose(const uint32_t *__restrict__ bytes, uint32_t *__restrict__ planes, unsigned long long *clk, int iters)
{
    const uint32_t lane = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t fold = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        uint32_t M[NP][32];
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            uint32_t w[8], P[NP];
#pragma unroll
            for (int d = 0; d < 8; ++d) w[d] = bytes[(p * 8 + d) * 64 + lane] ^ fold; // (every wave reads the same 64 KB: cache-resident -- the issue cost is what is measured)
            build_planes<NP>(w, P);
#pragma unroll
            for (int k = 0; k < NP; ++k) M[k][p] = P[k];
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) transpose32(M[k]);
        // (in a real kernel the planes would stay in registers / LDS: folded into one word here instead of 40 KB of stores per wave)
        uint32_t x = 0u;
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int i = 0; i < 32; ++i) x = bitop3<0x96>(x, M[k][i], (uint32_t)(i * NP + k));
        planes[blockIdx.x * 64 + lane] = x;
        fold = x & (it == 0x7FFFFFFF ? 1u : 0u);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

static uint32_t dp(const uint8_t *a, int la, const uint8_t *b, int lb)
{
    std::vector<uint32_t> row(lb + 1);
    for (int j = 0; j <= lb; ++j) row[j] = j;
    for (int i = 0; i < la; ++i) {
        uint32_t diag = row[0];
        row[0] = i + 1;
        for (int j = 0; j < lb; ++j) {
            const uint32_t v = std::min(std::min(row[j + 1] + 1, row[j] + 1), diag + (a[i] != b[j] ? 1u : 0u));
            diag = row[j + 1];
            row[j + 1] = v;
        }
    }
    return row[lb];
}

template <int L>
static void run_sliced(int cus, int waves_per_simd)
{
    const int blocks = cus * 4 * waves_per_simd, iters = 20;
    // one set of 2 048 pairs (the same for every block): random a-z strings, half of them near copies
    std::vector<uint8_t> A(2048 * L), Bs(2048 * L);
    srand(1234 + L);
    for (int p = 0; p < 2048; ++p)
        for (int i = 0; i < L; ++i) {
            A[p * L + i] = 'a' + rand() % 26;
            Bs[p * L + i] = (p & 1) && rand() % 8 ? A[p * L + i] : 'a' + rand() % 26;
        }
    std::vector<uint32_t> pa(L * NP * 64, 0), pb(L * NP * 64, 0);
    for (int lane = 0; lane < 64; ++lane)
        for (int bit = 0; bit < 32; ++bit)
            for (int i = 0; i < L; ++i)
                for (int k = 0; k < NP; ++k) {
                    const int p = lane * 32 + bit;
                    pa[(i * NP + k) * 64 + lane] |= (uint32_t)((A[p * L + i] >> k) & 1) << bit;
                    pb[(i * NP + k) * 64 + lane] |= (uint32_t)((Bs[p * L + i] >> k) & 1) << bit;
                }
    uint32_t *dpa, *dpb, *dd;
    unsigned long long *dc;
    hipMalloc(&dpa, pa.size() * 4); hipMalloc(&dpb, pb.size() * 4); hipMalloc(&dd, (size_t)blocks * 7 * 64 * 4); hipMalloc(&dc, blocks * 8);
    hipMemcpy(dpa, pa.data(), pa.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpb, pb.data(), pb.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sliced<L>, dim3(blocks), dim3(64), 0, 0, dpa, dpb, dd, dc, iters);
    hipDeviceSynchronize();
    std::vector<uint32_t> d(7 * 64);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < 2048; ++p) {
        int v = 0;
        for (int q = 0; q < 7; ++q) v |= ((d[q * 64 + p / 32] >> (p % 32)) & 1) << q;
        if (v & 64) v -= 128;
        const uint32_t got = (uint32_t)(L + v), exp = dp(&A[p * L], L, &Bs[p * L], L);
        bad += got != exp;
    }
    double cyc = 0;
    for (auto x : c) cyc += (double)x;
    cyc /= blocks; // s_memtime ticks (100 MHz * ... : the shader clock counter) per wave for `iters` sets of 2 048 pairs
    // a SIMD runs waves_per_simd waves at once: SIMD cycles per pair = wave cycles / (iters * 2048 * waves_per_simd)
    printf("sliced  L = %2d x %2d  %d waves/SIMD: %8.0f cycles per wave and 2 048 pairs -> %6.2f SIMD cycles per pair (%.4f per cell), %d of 2048 distances wrong\n",
           L, L, waves_per_simd, cyc / iters, cyc / iters / 2048.0 / waves_per_simd * 1.0, cyc / iters / 2048.0 / waves_per_simd / (L * L), bad);
    hipFree(dpa); hipFree(dpb); hipFree(dd); hipFree(dc);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int w : {2, 4}) {
        run_sliced<16>(cus, w);
        run_sliced<32>(cus, w);
    }
    for (int w : {2, 4}) {
        const int blocks = cus * 4 * w, iters = 10;
        uint32_t *db, *dp_;
        unsigned long long *dc;
        hipMalloc(&db, (size_t)blocks * 32 * 8 * 64 * 4); hipMalloc(&dp_, (size_t)blocks * 32 * NP * 64 * 4); hipMalloc(&dc, blocks * 8);
        hipMemset(db, 0x61, (size_t)blocks * 32 * 8 * 64 * 4);
        hipLaunchKernelGGL(k_transpose, dim3(blocks), dim3(64), 0, 0, db, dp_, dc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(blocks);
        hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (auto x : c) cyc += (double)x;
        cyc /= blocks;
        printf("transpose of one string's 32-byte windows, %d waves/SIMD: %8.0f cycles per wave and 2 048 windows -> %6.2f SIMD cycles per pair for BOTH strings\n",
               w, cyc / iters, 2.0 * cyc / iters / 2048.0 / w);
        hipFree(db); hipFree(dp_); hipFree(dc);
    }
    printf("(s_memtime counts at the shader clock; a SIMD's cycles per pair = a wave's cycles / pairs / the waves that share the SIMD.\n"
           " lev_myers32_snap: 12.7 SIMD cycles per pair on cfg2's lengths + 3.0 for its plane build, DESIGN 3.1a)\n");
    return 0;
}
