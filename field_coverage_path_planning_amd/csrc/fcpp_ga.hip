// fcpp_ga.hip -- the GA's evolution loop on the device (include/fcpp.h: fcpp_ga_evolve; GA:64-115, 183-268).
//
// One launch per generation (k_ga_generation: both roles below side by side; tours too long for that: two launches on two streams),
// no host round trip:
//   k_ga_pairs        one wavefront per pair of offspring: two tournaments (GA:183-196), order crossover of the winners
//                     (GA:212-242: the kept segment is marked in LDS, the donor's genes are ranked with ballots and
//                     scattered to (b + rank) mod n), swap mutation (GA:244-252), and the children's tour length in the
//                     reference's left-to-right order (GA:174-181), straight from LDS.  Rows that elitism overwrites
//                     (the last elite_size ones, GA:266) are not written.
//   k_ga_stats_elite  one workgroup: first-argmax and mean of the new fitness, best-so-far and convergence bookkeeping
//                     (GA:90-113), then the elites of this population (GA:254-268) into the tail of the next buffer.
// Every random decision is a pure function of (seed, generation, pair): Philox4x32-10, see include/fcpp.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "fcpp_ga.h"

namespace fcpp {

struct U4 { uint32_t w[4]; };

__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{ { c0, c1, c2, c3 } };
}

__device__ __forceinline__ double unit(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

__device__ __forceinline__ void two_positions(uint32_t r0, uint32_t r1, int n, int &i, int &j)
{
    i = (int)(r0 % (uint32_t)n);
    j = (int)(r1 % (uint32_t)(n - 1));
    if (j >= i) ++j;
}

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Order crossover of both children at once (GA:225-237): child c keeps parent c's genes [a, b) in place; its other positions, from b on
// (wrapping), take the OTHER parent's genes in the order they appear from position b on, skipping the genes already present.  The two
// children are independent: done side by side, each LDS round trip serves both.  pres0 / pres1: n zero bytes each on entry.
__device__ __forceinline__ void ox_children(const int32_t *P0, const int32_t *P1, int n, int a, int b, int32_t *C0, int32_t *C1,
                                            unsigned char *pres0, unsigned char *pres1, int lane)
{
    for (int i = a + lane; i < b; i += 64) {
        const int32_t g0 = P0[i], g1 = P1[i];
        C0[i] = g0; pres0[g0] = 1; C1[i] = g1; pres1[g1] = 1;
    }
    wsync();
    const unsigned long long below = (1ull << lane) - 1ull;
    int base0 = 0, base1 = 0;
    for (int q0 = 0; q0 < n; q0 += 64) {
        const int q = q0 + lane;
        int32_t e0 = 0, e1 = 0;
        bool f0 = false, f1 = false;
        if (q < n) {
            int src = b + q; if (src >= n) src -= n;
            e0 = P1[src]; e1 = P0[src];
            f0 = !pres0[e0]; f1 = !pres1[e1];
        }
        const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
        if (f0) { int pos = b + base0 + __popcll(m0 & below); if (pos >= n) pos -= n; C0[pos] = e0; }
        if (f1) { int pos = b + base1 + __popcll(m1 & below); if (pos >= n) pos -= n; C1[pos] = e1; }
        base0 += __popcll(m0); base1 += __popcll(m1);
    }
    wsync();
}

// wave-wide arg-max of (fitness, index) pairs in the order of the elitism: larger fitness first, among equals the larger index;
// lanes without a candidate pass idx < 0 (their fitness is ignored).  The maximum fitness goes through DPP moves (no index is
// carried along), one ballot finds who holds it -- normally one lane, whose index is read directly; several lanes with exactly the
// same fitness settle the index by a butterfly.  Returns the winning index (-1: no candidate) in every lane, its fitness in `wf`.
template <int CTRL>
__device__ __forceinline__ double ga_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ga_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int wave_argmax(double f, int idx, double &wf)
{
    double v = idx >= 0 ? f : 0.0;                       // fitness values are positive: 0 is "nothing"
    v = ga_max(v, ga_dpp<0xB1>(v));                      // quad_perm [1,0,3,2]
    v = ga_max(v, ga_dpp<0x4E>(v));                      // quad_perm [2,3,0,1]
    v = ga_max(v, ga_dpp<0x141>(v));                     // row_half_mirror
    v = ga_max(v, ga_dpp<0x140>(v));                     // row_mirror
    v = ga_max(v, ga_dpp<0x142>(v));                     // row_bcast:15
    v = ga_max(v, ga_dpp<0x143>(v));                     // row_bcast:31: lane 63 holds the maximum
    wf = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
    const unsigned long long m = __ballot(idx >= 0 && f == wf);
    if (m == 0ull) return -1;
    if ((m & (m - 1ull)) == 0ull) return __builtin_amdgcn_readlane(idx, (int)__ffsll((long long)m) - 1);      // (a scalar lane index)
    int wi = (idx >= 0 && f == wf) ? idx : -1;           // ties: the larger index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wi = max(wi, __shfl_xor(wi, o));
    return wi;
}

// Diagnostic build only (-DFCPP_DIAG_GA: `make diag-ga`, tools/diag_ga.py; never shipped): 10 ns time stamps of the phases of ONE pair's
// wavefront (pair 1000 of generation 250) and of the two bookkeeping workgroups of the same launch.
#ifdef FCPP_DIAG_GA
__device__ unsigned long long g_ga_stamps[48];
#define GSTAMP(k) do { if (gen == 250 && pair == 1000 && lane == 0) g_ga_stamps[k] = wall_clock64(); } while (0)
#define BSTAMP(k) do { if (gen == 249 && threadIdx.x == 0) g_ga_stamps[k] = wall_clock64(); } while (0)
#else
#define GSTAMP(k) do { } while (0)
#define BSTAMP(k) do { } while (0)
#endif

// int32 words of a wavefront's LDS slice before its 2 x 64 tournament candidates: 4 n genes + 2 n presence bytes (one set per child),
// rounded to 16 bytes
#define GA_PAIR_LDS_HEAD(n) ((((size_t)(n) * 4 * sizeof(int32_t) + (size_t)(2 * (n)) + 15) / 16) * 4)
// one wavefront, one pair of offspring.  lds: 4 n genes + n presence bytes + 128 candidate indices of this wavefront, s_w: two ints of it
__device__ __forceinline__ void ga_pair(int lane, int pair, int32_t *lds, int *s_w, int n, int pop, const double *__restrict__ D,
                                        const int32_t *__restrict__ cur, const double *__restrict__ cur_fit, int32_t *__restrict__ nxt,
                                        double *__restrict__ nxt_fit, double *__restrict__ nxt_dist, const fcpp_ga_config &cfg, int gen, int conv)
{
    int32_t *const P[2] = { lds, lds + n }, *const Cc[2] = { lds + 2 * n, lds + 3 * n };
    unsigned char *const pres0 = reinterpret_cast<unsigned char *>(lds + 4 * n), *const pres1 = pres0 + n;
    const uint32_t k0 = (uint32_t)cfg.seed, k1 = (uint32_t)(cfg.seed >> 32);
    GSTAMP(0);
    // (the crossover's presence marks are cleared now, long before they are needed; the slice is a multiple of 16 bytes)
    for (int w = lane; w < (2 * n + 3) / 4; w += 64) lds[4 * n + w] = 0;
    // The two tournaments (GA:189-194): k distinct candidates each, the FIRST maximum wins (np.argmax).
    // Fast path (k <= 8): every random number of the pair comes from ONE Philox evaluation of the wavefront -- lanes 0-15 the first 16
    // draws of tournament 1 (draw j = word j & 3 of block j >> 2 of its stream), lanes 16-31 those of tournament 2, lanes 32 / 33 / 34
    // the crossover's block and the two mutations' -- instead of five evaluations one after the other.  A draw is a candidate unless an
    // EARLIER draw of its tournament has the same value (which is what the sequential rejection of repeats amounts to), the first k of them
    // play; the candidates' fitness is fetched by their own lanes, and the winner is the first lane holding the maximum.  Fewer than k
    // distinct values among 16 draws (tiny populations): the sequential path below.
    int w0 = -1, w1 = -1;
    U4 X = { { 0, 0, 0, 0 } }, M = { { 0, 0, 0, 0 } };
    bool fast = cfg.tournament_size <= 8;
    if (fast) {
        const int k = cfg.tournament_size, grp = lane >> 4, j = lane & 15;
        const U4 blk = philox((uint32_t)gen, (uint32_t)pair, grp == 0 ? 1u : (grp == 1 ? 2u : 3u + (uint32_t)j), grp < 2 ? (uint32_t)(j >> 2) : 0u, k0, k1);
        const uint32_t word = (j & 3) == 0 ? blk.w[0] : ((j & 3) == 1 ? blk.w[1] : ((j & 3) == 2 ? blk.w[2] : blk.w[3]));
        const int c = (int)(word % (uint32_t)pop);
        bool dup = false;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int v0 = __builtin_amdgcn_readlane(c, i), v1 = __builtin_amdgcn_readlane(c, 16 + i);
            dup |= j > i && c == (grp == 0 ? v0 : v1);
        }
        const bool first = grp < 2 && !dup;
        const unsigned long long fm = __ballot(first);
        const unsigned g0 = (unsigned)(fm & 0xffffull), g1 = (unsigned)((fm >> 16) & 0xffffull);
        if (__popc(g0) >= k && __popc(g1) >= k) {
            const unsigned mine = grp == 0 ? g0 : g1;
            const bool sel = first && __popc(mine & ((1u << j) - 1u)) < k;
            const double f = sel ? cur_fit[c] : 0.0;                 // (both tournaments' loads in flight together)
            double m0, m1;
            (void)wave_argmax(f, (sel && grp == 0) ? lane : -1, m0);  // (only the maximum is used: the FIRST lane holding it wins)
            (void)wave_argmax(f, (sel && grp == 1) ? lane : -1, m1);
            const unsigned long long b0 = __ballot(sel && grp == 0 && f == m0), b1 = __ballot(sel && grp == 1 && f == m1);
            w0 = __builtin_amdgcn_readlane(c, (int)__ffsll((long long)b0) - 1);
            w1 = __builtin_amdgcn_readlane(c, (int)__ffsll((long long)b1) - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                X.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)blk.w[q], 32);
                const uint32_t ma = (uint32_t)__builtin_amdgcn_readlane((int)blk.w[q], 33), mb = (uint32_t)__builtin_amdgcn_readlane((int)blk.w[q], 34);
                M.w[q] = lane == 0 ? ma : mb;
            }
        } else {
            fast = false;
        }
    }
    if (!fast) {
        // Lanes 0 and 1 draw the candidates (the stream of draws and the rejection of repeats are sequential) into LDS -- a per-lane
        // array indexed at run time would live in scratch memory, a memory round trip per look-up -- then ALL lanes fetch the candidates'
        // fitness at once.
        int32_t *const s_cand = lds + GA_PAIR_LDS_HEAD(n);          // 2 x 64 candidate indices of this wavefront
        if (lane < 2) {
            int nc = 0;
            U4 blk = { { 0, 0, 0, 0 } };
            for (uint32_t j = 0; nc < cfg.tournament_size; ++j) {
                if ((j & 3u) == 0u) blk = philox((uint32_t)gen, (uint32_t)pair, (uint32_t)(1 + lane), j >> 2, k0, k1);
                const uint32_t word = (j & 3u) == 0u ? blk.w[0] : ((j & 3u) == 1u ? blk.w[1] : ((j & 3u) == 2u ? blk.w[2] : blk.w[3]));
                const int c = (int)(word % (uint32_t)pop);
                bool dup = false;
                for (int q = 0; q < nc; ++q) dup |= s_cand[64 * lane + q] == c;
                if (!dup) { s_cand[64 * lane + nc] = c; ++nc; }
            }
        }
        wsync();
        {
            const int k = cfg.tournament_size;
            const int c0 = lane < k ? s_cand[lane] : -1, c1 = lane < k ? s_cand[64 + lane] : -1;
            const double f0 = c0 >= 0 ? cur_fit[c0] : 0.0, f1 = c1 >= 0 ? cur_fit[c1] : 0.0;      // (both tournaments' loads in flight together)
            double m0, m1;
            (void)wave_argmax(f0, c0 >= 0 ? lane : -1, m0);
            (void)wave_argmax(f1, c1 >= 0 ? lane : -1, m1);
            const unsigned long long b0 = __ballot(c0 >= 0 && f0 == m0), b1 = __ballot(c1 >= 0 && f1 == m1);
            if (lane == 0) { s_w[0] = s_cand[__ffsll((long long)b0) - 1]; s_w[1] = s_cand[64 + __ffsll((long long)b1) - 1]; }
        }
        wsync();
        w0 = __builtin_amdgcn_readfirstlane(s_w[0]); w1 = __builtin_amdgcn_readfirstlane(s_w[1]);
        X = philox((uint32_t)gen, (uint32_t)pair, 3u, 0u, k0, k1);
        if (lane < 2) M = philox((uint32_t)gen, (uint32_t)pair, (uint32_t)(4 + lane), 0u, k0, k1);
    }
    GSTAMP(1);
    const int32_t *p1g = cur + (int64_t)w0 * n, *p2g = cur + (int64_t)w1 * n;
    for (int i = lane; i < n; i += 64) { P[0][i] = p1g[i]; P[1][i] = p2g[i]; }
    wsync();
    GSTAMP(2);
    if (unit(X.w[0], X.w[1]) < cfg.crossover_rate) {      // GA:207
        int i, j;
        two_positions(X.w[2], X.w[3], n, i, j);
        const int a = min(i, j), b = max(i, j);
        ox_children(P[0], P[1], n, a, b, Cc[0], Cc[1], pres0, pres1, lane);
    } else {
        for (int i = lane; i < n; i += 64) { Cc[0][i] = P[0][i]; Cc[1][i] = P[1][i]; }
        wsync();
    }
    GSTAMP(3);
    if (lane < 2) {      // GA:246-250
        if (unit(M.w[0], M.w[1]) < cfg.mutation_rate) {
            int i, j;
            two_positions(M.w[2], M.w[3], n, i, j);
            const int32_t t = Cc[lane][i]; Cc[lane][i] = Cc[lane][j]; Cc[lane][j] = t;
        }
    }
    wsync();
    // The children's rows and tour lengths (GA:174-181).  The matrix entries of BOTH children are asked for first (tours of up to 256
    // nodes: eight loads in flight instead of eight round trips), the rows are stored, and the terms go to LDS -- over the parents' and
    // children's genes, which nobody needs any more -- where lane c adds child c's terms left to right, as the reference's loop adds
    // them: one LDS read and one addition per term, both children by the same instructions.  (Handing lane l's term to the sum through
    // v_readlane cost four instructions per term and child: 55 cycles per step of the two chains, 3.3 of a generation's 15 us.)
    const int nchunk = (n + 63) >> 6;
    const bool rowok[2] = { 2 * pair < pop - cfg.elite_size, 2 * pair + 1 < pop - cfg.elite_size };     // an elite takes the other rows (GA:266)
    double *const S = reinterpret_cast<double *>(lds);             // 2 n doubles = the 4 n gene words
    auto sum_left_to_right = [&](const double *t, int m) -> double {
        double total = 0.0;
        int l = 0;
        if ((reinterpret_cast<uintptr_t>(t) & 15) == 0) {
            // (16-byte LDS reads: half as many LDS instructions -- the eight pair wavefronts of a compute unit share one LDS pipe, and a read
            // costs it the same whether two lanes are active or sixty-four)
            const double2 *t2 = reinterpret_cast<const double2 *>(t);
            for (; l + 16 <= m; l += 16) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t2[(l >> 1) + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) { total += v[u].x; total += v[u].y; }
            }
        }
        for (; l + 16 <= m; l += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = t[l + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) total += v[u];
        }
        for (; l < m; ++l) total += t[l];
        return total;
    };
    GSTAMP(4);
    if (conv) return;                      // (the run has converged: nothing is written)
    if (nchunk <= 4) {
        double dd[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int32_t *ch = Cc[c];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = 64 * b + lane;
                dd[c][b] = (rowok[c] && k < n) ? D[(int64_t)ch[k] * n + ch[k + 1 == n ? 0 : k + 1]] : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (rowok[c]) for (int i = lane; i < n; i += 64) nxt[(int64_t)(2 * pair + c) * n + i] = Cc[c][i];
        wsync();                           // every lane has read its genes
        GSTAMP(5);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int b = 0; b < 4; ++b) { const int k = 64 * b + lane; if (k < n) S[c * n + k] = dd[c][b]; }
        wsync();
        GSTAMP(6);
        if (lane < 2 && (lane == 0 ? rowok[0] : rowok[1])) {
            const double t = sum_left_to_right(S + lane * n, n);
            nxt_dist[2 * pair + lane] = t; nxt_fit[2 * pair + lane] = 1.0 / (t + 1e-6);
        }
        GSTAMP(7);
        return;
    }
    // long tours: child 0's terms go over the parents' genes, child 1's over them again once child 0's sum is taken (its terms would lie
    // over the children's genes, which its own look-ups still read)
    for (int c = 0; c < 2; ++c) {
        const int row = 2 * pair + c;
        if (!rowok[c]) continue;
        const int32_t *ch = Cc[c];
        for (int i = lane; i < n; i += 64) nxt[(int64_t)row * n + i] = ch[i];
        double *const T = reinterpret_cast<double *>(lds);         // n doubles = P[0] | P[1]
        for (int base = 0; base < n; base += 64) {
            const int k = base + lane;
            if (k < n) T[k] = D[(int64_t)ch[k] * n + ch[k + 1 == n ? 0 : k + 1]];
        }
        wsync();
        if (lane == 0) {
            const double t = sum_left_to_right(T, n);
            nxt_dist[row] = t; nxt_fit[row] = 1.0 / (t + 1e-6);
        }
        wsync();                           // (the sum is taken before the next child's terms overwrite it)
    }
}

__global__ __launch_bounds__(64) void k_ga_pairs(int n, int pop, const double *__restrict__ D, const int32_t *__restrict__ cur,
                                                 const double *__restrict__ cur_fit, int32_t *__restrict__ nxt,
                                                 double *__restrict__ nxt_fit, double *__restrict__ nxt_dist, fcpp_ga_config cfg,
                                                 int gen, const GaState *__restrict__ state)
{
    extern __shared__ int32_t lds[];                    // 4 n genes + n presence bytes (sized by the launcher: small tours -> many waves per CU)
    __shared__ int s_w[2];
    // (the flag is asked for first and looked at last, before anything is written: its round trip runs beside the pair's own loads)
    const int conv = __atomic_load_n(&state->converged, __ATOMIC_RELAXED);
    ga_pair(threadIdx.x, blockIdx.x, lds, s_w, n, pop, D, cur, cur_fit, nxt, nxt_fit, nxt_dist, cfg, gen, conv);
}

static constexpr int GA_LDS_POP = 6144;      // 48 KiB of fitness values cached in LDS
static constexpr int SB = 1024, SW = SB / 64; // the bookkeeping kernel: one workgroup
static constexpr int GA_ELITE_CAND = 256;     // candidates the short elite selection ranks

// block-wide reduction of (value, index) pairs with a caller-supplied "a is better than b": shuffles inside the waves, one
// LDS exchange across them.  Every thread returns the winner.
template <class Better>
__device__ __forceinline__ void block_best(double &f, int &i, double *s_f, int *s_i, Better better)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double of = __shfl_xor(f, o);
        const int oi = __shfl_xor(i, o);
        if (better(of, oi, f, i)) { f = of; i = oi; }
    }
    __syncthreads();                 // (previous users of s_f / s_i are done)
    if (lane == 0) { s_f[wave] = f; s_i[wave] = i; }
    __syncthreads();
    f = s_f[0]; i = s_i[0];
#pragma unroll
    for (int w = 1; w < SW; ++w) if (better(s_f[w], s_i[w], f, i)) { f = s_f[w]; i = s_i[w]; }
}

// one workgroup of SB threads.  s_fit: pop doubles of LDS when pop <= GA_LDS_POP.  ROLE 0: statistics and best-so-far bookkeeping, then
// the elites (one workgroup does both: k_ga_stats_elite); 1: the bookkeeping alone; 2: the elites alone -- the two are independent chains
// over the same fitness values (the elites of a generation in which convergence is found go, like its children, to the buffer that is
// not the result), so k_ga_generation gives each a workgroup of its own and a generation costs the longer one, not their sum.
template <int ROLE>
__device__ __forceinline__ void ga_stats_elite(double *s_fit, int n, int pop, const int32_t *__restrict__ cur, const double *__restrict__ cur_fit,
                                               const double *__restrict__ cur_dist, int32_t *__restrict__ nxt,
                                               double *__restrict__ nxt_fit, double *__restrict__ nxt_dist, const fcpp_ga_config &cfg,
                                               int gen, GaState *__restrict__ state, int32_t *__restrict__ best_route,
                                               double *__restrict__ hist, int conv)
{
    __shared__ double s_f[SW], s_sum[SB];
    __shared__ int s_i[SW], s_pick[64];
    __shared__ int s_copy, s_stop;
    const int tid = threadIdx.x;
    const bool cached = pop <= GA_LDS_POP;
    const double *__restrict__ fitv = cached ? s_fit : cur_fit;
    const double thr_prev = ROLE != 1 ? state->elite_thr : 0.0;      // (asked for first, used after the fitness values have arrived)
    // first argmax (np.argmax, GA:66 / GA:91) and the mean (GA:107) in a fixed order: SB strided partial sums, then a binary tree
    double bf = -1.0, acc = 0.0;
    int bi = 0x7fffffff;
    for (int i = tid; i < pop; i += SB) {
        const double f = cur_fit[i];
        if (cached && ROLE != 1) s_fit[i] = f;
        acc += f;
        if (f > bf || (f == bf && i < bi)) { bf = f; bi = i; }
    }
    if (ROLE == 2) {
        if (gen + 1 >= cfg.max_generations) return;
        __syncthreads();                  // the fitness values are in LDS
    }
    if (ROLE != 2) {
    s_sum[tid] = acc;
    __syncthreads();
    for (int o = SB / 2; o >= 64; o >>= 1) {
        if (tid < o) s_sum[tid] += s_sum[tid + o];
        __syncthreads();
    }
    double total = 0.0;
    if (tid < 64) {                       // the last six levels of the same tree inside one wave
        total = s_sum[tid];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o);
    }
    block_best(bf, bi, s_f, s_i, [](double af, int ai, double cf, int ci) { return af > cf || (af == cf && ai < ci); });
    if (tid == 0) {
        const double f = bf;
        const int i = bi;
        int copy = -1, stop = 0;
        if (gen < 0) {                                   // GA:66-70
            state->best_fit = f; state->best_dist = cur_dist[i]; state->gwi = 0; state->generations = 0;
            copy = i;
        } else {
            if (f > state->best_fit) { state->best_fit = f; state->best_dist = cur_dist[i]; state->gwi = 0; copy = i; }   // GA:94-98
            else state->gwi += 1;
            if (hist) { hist[gen] = state->best_fit; hist[cfg.max_generations + gen] = total / (double)pop; }             // GA:106-107
            state->generations = gen + 1;
            if (state->gwi >= cfg.convergence_threshold) { state->converged = 1; stop = 1; }                              // GA:110-113
        }
        s_copy = copy; s_stop = stop;
        if (state->mirror)
            __hip_atomic_store(state->mirror, ((unsigned long long)(unsigned)state->converged << 32) | (unsigned long long)(unsigned)state->generations,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (s_copy >= 0)
        for (int k = tid; k < n; k += SB) best_route[k] = cur[(int64_t)s_copy * n + k];
    if (ROLE == 1 || s_stop || gen + 1 >= cfg.max_generations) return;
    }
    // elites of this population for the next generation (GA:254-268): the t-th best in the order (fitness, index) goes to row
    // pop - 1 - t.
    auto better = [](double af, int ai, double cf, int ci) { return ai >= 0 && (ci < 0 || af > cf || (af == cf && ai > ci)); };
    if (cached && cfg.elite_size >= 1 && cfg.elite_size <= 64) {
        // The short way: the elites of the previous generation were carried over into the last E rows, so E members reach the fitness
        // of its last elite (thr_prev), and a member that does not exceed it loses to every one of them (equal fitness: the larger
        // index wins, and theirs are the largest).  The candidates are therefore the last E rows plus the members strictly above
        // thr_prev -- normally a few children; in a population that has collapsed into copies of one tour, none.  They are gathered in
        // LDS (in any order: the elitism's order is total), every candidate counts the candidates that precede it, and rank t < E is
        // pick t.  More than GA_ELITE_CAND candidates (the first selection of a run: thr_prev = 0; many copies of the better elites):
        // the selection below.
        __shared__ double q_f[GA_ELITE_CAND];
        __shared__ int q_i[GA_ELITE_CAND];
        __shared__ int q_n;
        const int E = cfg.elite_size;
        __shared__ int q_rank[GA_ELITE_CAND];
        if (tid == 0) q_n = 0;
        if (tid < GA_ELITE_CAND) q_rank[tid] = 0;
        __syncthreads();
        // (the candidates of a wavefront take their places together: a ballot, one addition to the counter per wavefront -- a candidate a
        // lane, each with an atomic of its own on the one counter, was a microsecond of serialised LDS atomics)
        for (int i0 = 0; i0 < pop; i0 += SB) {
            const int i = i0 + tid;
            const double f = i < pop ? s_fit[i] : 0.0;
            const bool is_c = i < pop && (f > thr_prev || i >= pop - E);
            const unsigned long long m = __ballot(is_c);
            if (m != 0ull) {
                const int lane = tid & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&q_n, __popcll(m));
                base = __shfl(base, 0);
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (is_c && pos < GA_ELITE_CAND) { q_f[pos] = f; q_i[pos] = i; }
            }
        }
        __syncthreads();
        const int C = q_n;
        if (C >= E && C <= GA_ELITE_CAND) {
            // every candidate's rank = the candidates that precede it: the C x C comparisons spread over all 1024 threads (thread t: candidate
            // t % Cp against every (SB / Cp)-th other), the partial counts met in LDS -- one candidate per thread walking all C others was
            // C dependent LDS round trips, 2.5 of the role's 9.3 us (profiles/r05_ga_phases.txt)
            int Cp = 1;
            while (Cp < C) Cp <<= 1;                      // (a power of two <= 256: SB / Cp threads per candidate)
            const int c = tid & (Cp - 1), part = tid / Cp, nparts = SB / Cp;
            if (c < C) {
                const double f = q_f[c];
                const int i = q_i[c];
                int cnt = 0;
                for (int u = part; u < C; u += nparts) {
                    const double uf = q_f[u];
                    const int ui = q_i[u];
                    cnt += (uf > f || (uf == f && ui > i)) ? 1 : 0;
                }
                if (cnt) atomicAdd(&q_rank[c], cnt);
            }
            __syncthreads();
            if (tid < C) {
                const int rank = q_rank[tid];
                if (rank < E) s_pick[rank] = q_i[tid];
                if (rank == E - 1 && !conv) state->elite_thr = q_f[tid];
            }
            __syncthreads();
            if (conv) return;
            for (int q = tid; q < E * n; q += SB) {
                const int t = q / n, k = q - t * n;
                nxt[(int64_t)(pop - 1 - t) * n + k] = cur[(int64_t)s_pick[t] * n + k];
            }
            if (tid < E) { const int src = s_pick[tid], row = pop - 1 - tid; nxt_fit[row] = fitv[src]; nxt_dist[row] = cur_dist[src]; }
            return;
        }
    }
    if (cached && cfg.elite_size <= 64) {
        // Up to 64 elites, the population in LDS: every wavefront lists the best of ITS 1/16 of the population -- rounds of a wave-wide
        // arg-max, all wavefronts at once -- and the lists are merged by rank.  The global top elite_size are among the per-wave ones, and the order (fitness, then the larger
        // index) is total, so the picks are those of the sequential definition; two dependent phases instead of elite_size.
        __shared__ double c_f[SW * 64];
        __shared__ int c_i[SW * 64], s_rank[SW * 64];
        constexpr int KPT = GA_LDS_POP / SB;
        const int lane = tid & 63, wave = tid >> 6, E = cfg.elite_size;
        double ef[KPT];
        unsigned taken = 0;
        // (key i belongs to wavefront i % 16: the previous generation's elites, which sit in consecutive rows and are likely to be picked
        // again, spread over all wavefronts instead of filling one wavefront's list)
        static_assert(SW == 16, "key ownership below");
#pragma unroll
        for (int k = 0; k < KPT; ++k) { const int i = ((k * 64 + lane) << 4) | wave; ef[k] = i < pop ? s_fit[i] : -1.0; }
        // The wavefronts' lists are first built ELITE_FIRST deep only: the global top E lie in the union of the per-wave top R as long as
        // no wavefront holds R of them, which the merge itself shows -- a wavefront whose R-th candidate is picked may hold more, and
        // only then are the lists continued to E and merged again (a wavefront holds E / 16 of the elites on average).
        constexpr int ELITE_FIRST = 6;
        int done = 0;
        for (int depth = min(E, ELITE_FIRST);; depth = E) {
            for (int t = done; t < depth; ++t) {
                double f = -1.0;
                int idx = -1, kk = 0;
#pragma unroll
                for (int k = 0; k < KPT; ++k) {
                    const int i = ((k * 64 + lane) << 4) | wave;
                    if (i < pop && !((taken >> k) & 1u) && better(ef[k], i, f, idx)) { f = ef[k]; idx = i; kk = k; }
                }
                double wf;
                const int wi = wave_argmax(f, idx, wf);
                if (idx >= 0 && wi == idx) taken |= 1u << kk;         // its owner retires the pick
                if (lane == 0) { c_f[wave * 64 + t] = wi >= 0 ? wf : -1.0; c_i[wave * 64 + t] = wi; }
            }
            done = depth;
            __syncthreads();
            // Merge by rank: every wavefront's candidates are in the elitism's order (best first, exhausted slots last), which is a
            // strict total order; thread (w, t) owns candidate t of wavefront w, and its global rank is t plus, for every other
            // wavefront, the number of that list's candidates that beat it -- a binary search per list, all 16 x depth candidates at
            // once (the 16 lists used to be merged by one wavefront in E dependent rounds: 22 of the kernel's 40 us).
            // (one task per candidate and OTHER list, spread over all threads; the partial ranks meet in LDS)
            if (lane < depth) s_rank[wave * 64 + lane] = lane;
            __syncthreads();
            for (int q = tid; q < SW * depth * SW; q += SB) {
                const int c = q / SW, w2 = q - c * SW, cw = c / depth, ct = c - cw * depth;
                const int ci0 = c_i[cw * 64 + ct];
                if (w2 == cw || ci0 < 0) continue;
                const double cf0 = c_f[cw * 64 + ct];
                int lo = 0, hi = depth;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (better(c_f[w2 * 64 + mid], c_i[w2 * 64 + mid], cf0, ci0)) lo = mid + 1; else hi = mid;
                }
                if (lo) atomicAdd(&s_rank[cw * 64 + ct], lo);
            }
            __syncthreads();
            const int myi = lane < depth ? c_i[wave * 64 + lane] : -1;
            int deeper = 0;
            if (myi >= 0) {
                const int rank = s_rank[wave * 64 + lane];
                if (rank < E) s_pick[rank] = myi;
                deeper = depth < E && lane == depth - 1 && rank < E;
            }
            if (!__syncthreads_or(deeper)) break;
        }
        if (conv) return;
        if (tid == 0 && E >= 1) state->elite_thr = fitv[s_pick[E - 1]];
        for (int q = tid; q < E * n; q += SB) {
            const int t = q / n, k = q - t * n;
            nxt[(int64_t)(pop - 1 - t) * n + k] = cur[(int64_t)s_pick[t] * n + k];
        }
        if (tid < E) { const int src = s_pick[tid], row = pop - 1 - tid; nxt_fit[row] = fitv[src]; nxt_dist[row] = cur_dist[src]; }
        return;
    }
    // general case: each pick is the maximum among the entries below the previous pick (elite_size dependent block reductions)
    double pf = 0.0;
    int pi = 0;
    for (int t0 = 0; t0 < cfg.elite_size; t0 += 64) {          // picks in batches of 64, their rows copied together
        const int nt = min(64, cfg.elite_size - t0);
        for (int t = t0; t < t0 + nt; ++t) {
            double f = -1.0;
            int idx = -1;
            for (int i = tid; i < pop; i += SB) {
                const double v = fitv[i];
                if (t > 0 && !(v < pf || (v == pf && i < pi))) continue;
                if (idx < 0 || v > f || (v == f && i > idx)) { f = v; idx = i; }
            }
            block_best(f, idx, s_f, s_i, better);
            pf = f; pi = idx;
            if (tid == 0) s_pick[t - t0] = idx;
        }
        __syncthreads();
        if (conv) return;
        for (int q = tid; q < nt * n; q += SB) {
            const int t = q / n, k = q - t * n;
            nxt[(int64_t)(pop - 1 - (t0 + t)) * n + k] = cur[(int64_t)s_pick[t] * n + k];
        }
        if (tid < nt) { const int src = s_pick[tid], row = pop - 1 - (t0 + tid); nxt_fit[row] = fitv[src]; nxt_dist[row] = cur_dist[src]; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SB) void k_ga_stats_elite(int n, int pop, const int32_t *__restrict__ cur, const double *__restrict__ cur_fit,
                                                       const double *__restrict__ cur_dist, int32_t *__restrict__ nxt,
                                                       double *__restrict__ nxt_fit, double *__restrict__ nxt_dist, fcpp_ga_config cfg,
                                                       int gen, GaState *__restrict__ state, int32_t *__restrict__ best_route,
                                                       double *__restrict__ hist)
{
    extern __shared__ double s_fit_dyn[];  // the population's fitness, read once (pop <= GA_LDS_POP; otherwise re-read from global)
    if (state->converged) return;
    ga_stats_elite<0>(s_fit_dyn, n, pop, cur, cur_fit, cur_dist, nxt, nxt_fit, nxt_dist, cfg, gen, state, best_route, hist, 0);
}

// One generation in ONE launch (small tours): population `cur` -> its statistics and best-so-far bookkeeping (generation index
// gen - 1, workgroup 0), its elites into the last rows of `nxt` (workgroup 1), and its children into the other rows
// (generation index gen; GA_PAIRS_PER_WG wavefronts = pairs per workgroup).  The two roles read the same population and write
// disjoint rows, so they need no order between them; a generation then costs the longer role, not the sum of two launches.
// stats_gen == -2: no bookkeeping role (never used); pairs_gen < 0: no children (the final population's statistics).
static constexpr int GA_PAIRS_PER_WG = 8;       // of the workgroup's 16 wavefronts (the others leave at once): two pair wavefronts per SIMD -- with sixteen pairs per
                                                // workgroup half the chip idled while four wavefronts shared every SIMD of the other half
// Round 5b: ALL workgroups of a launch resident at once.  The kernel holds 98 vector registers -- four wavefronts per SIMD, so a compute unit takes
// ONE workgroup of 1024 threads, and a pair workgroup's eight idle wavefronts free nothing a sixteen-wavefront workgroup could use.  2048 pairs as 256
// workgroups of 8 + the two single roles = 258 workgroups on 256 compute units: the last two waited for a compute unit to drain, and the launch took two
// roles' time (11.5 us for roles of at most 5.2).  The pairs are now spread over (compute units - 2) workgroups, 8 or 9 of them each (at most
// GA_PAIRS_MAX; more pairs than that: 8 per workgroup and as many workgroups as it takes, as before): workgroup b takes the pairs
// [b base + min(b, extra), ...) with base = pairs / workgroups, extra = pairs % workgroups.
static constexpr int GA_PAIRS_MAX = 16;
__global__ __launch_bounds__(SB) void k_ga_generation(int n, int pop, const double *__restrict__ D, const int32_t *__restrict__ cur,
                                                      const double *__restrict__ cur_fit, const double *__restrict__ cur_dist,
                                                      int32_t *__restrict__ nxt, double *__restrict__ nxt_fit, double *__restrict__ nxt_dist,
                                                      fcpp_ga_config cfg, int gen, int pair_lds_ints, int pairs_base, int pairs_extra,
                                                      GaState *__restrict__ state, int32_t *__restrict__ best_route, double *__restrict__ hist)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_w[GA_PAIRS_MAX][2];
    // (the children of the generation in which convergence is found go to the buffer that is not the result; the pairs and the elites
    // ask for the flag first and look at it last, before anything is written: its round trip runs beside their own loads)
    const int conv = __atomic_load_n(&state->converged, __ATOMIC_RELAXED);
    if (blockIdx.x == 0) {
        if (conv) return;
        { const int gen_ = gen; (void)gen_; }
        BSTAMP(16);
        ga_stats_elite<1>(dyn_lds, n, pop, cur, cur_fit, cur_dist, nxt, nxt_fit, nxt_dist, cfg, gen - 1, state, best_route, hist, 0);
        BSTAMP(17);
        return;
    }
    if (blockIdx.x == 1) {
        BSTAMP(18);
        ga_stats_elite<2>(dyn_lds, n, pop, cur, cur_fit, cur_dist, nxt, nxt_fit, nxt_dist, cfg, gen - 1, state, best_route, hist, conv);
        BSTAMP(19);
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = (int)blockIdx.x - 2;
    const int mine = pairs_base + (b < pairs_extra ? 1 : 0);                       // this workgroup's pairs, a wavefront each
    const int pair = b * pairs_base + (b < pairs_extra ? b : pairs_extra) + wave;
    if (wave >= mine || pair >= pop / 2 || gen >= cfg.max_generations) return;
    GSTAMP(8);
    ga_pair(lane, pair, reinterpret_cast<int32_t *>(dyn_lds) + (size_t)wave * pair_lds_ints, s_w[wave], n, pop, D, cur, cur_fit, nxt, nxt_fit,
            nxt_dist, cfg, gen, conv);
}

// precondition of fcpp_ga_evolve: every row of `routes` is a permutation of 0 .. n-1.  One wavefront per chromosome marks its genes
// in LDS; a gene out of range or seen twice raises the flag.
__global__ __launch_bounds__(64) void k_ga_check_perm(int n, int pop, const int32_t *__restrict__ routes, int32_t *__restrict__ bad)
{
    __shared__ unsigned char seen[GA_MAX_NODES];
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= pop) return;
    for (int g = lane; g < n; g += 64) seen[g] = 0;
    wsync();
    bool wrong = false;
    for (int k0 = 0; k0 < n; k0 += 64) {       // 64 genes at a time: duplicates inside one batch are caught by the count below
        const int k = k0 + lane;
        if (k < n) {
            const int32_t g = routes[(int64_t)ch * n + k];
            if ((unsigned)g >= (unsigned)n) wrong = true;
            else seen[g] = 1;
        }
    }
    wsync();
    int cnt = 0;
    for (int g = lane; g < n; g += 64) cnt += seen[g];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (__ballot(wrong) != 0ull || cnt != n) { if (lane == 0) atomicOr(bad, 1); }
}

int launch_ga_check_perm(hipStream_t st, int n, int pop, const int32_t *routes, int32_t *bad)
{
    hipLaunchKernelGGL(k_ga_check_perm, dim3((unsigned)pop), dim3(64), 0, st, n, pop, routes, bad);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_ga_pairs(hipStream_t st, int n, int pop, const double *D, const int32_t *cur, const double *cur_fit, int32_t *nxt,
                    double *nxt_fit, double *nxt_dist, const fcpp_ga_config &cfg, int gen, const GaState *state)
{
    const size_t lds = (GA_PAIR_LDS_HEAD(n) + 128) * sizeof(int32_t);
    hipLaunchKernelGGL(k_ga_pairs, dim3((unsigned)(pop / 2)), dim3(64), lds, st, n, pop, D, cur, cur_fit, nxt, nxt_fit, nxt_dist, cfg, gen,
                       state);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_ga_stats_elite(hipStream_t st, int n, int pop, const int32_t *cur, const double *cur_fit, const double *cur_dist, int32_t *nxt,
                          double *nxt_fit, double *nxt_dist, const fcpp_ga_config &cfg, int gen, GaState *state, int32_t *best_route,
                          double *hist)
{
    const size_t lds = pop <= GA_LDS_POP ? (size_t)pop * sizeof(double) : 0;
    hipLaunchKernelGGL(k_ga_stats_elite, dim3(1), dim3(SB), lds, st, n, pop, cur, cur_fit, cur_dist, nxt, nxt_fit, nxt_dist, cfg, gen, state,
                       best_route, hist);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// lds ints of one pair: 4 n genes + n presence bytes, rounded to 16 bytes
static inline size_t ga_pair_lds_ints(int n) { return GA_PAIR_LDS_HEAD(n) + 128; }

// how a launch spreads pop / 2 pairs over its workgroups: -> pair workgroups; base / extra as k_ga_generation reads them; per_wg = the most pairs of one
static unsigned ga_pair_groups(int n, int pop, int &base, int &extra, int &per_wg)
{
    static const int n_cu = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cu = 0; }
        return cu;
    }();
    const int pairs = pop / 2;
    int groups = n_cu > 2 ? n_cu - 2 : 0;
    if (groups > pairs) groups = pairs;
    const int most = groups > 0 ? (pairs + groups - 1) / groups : 0;
    if (groups <= 0 || most > GA_PAIRS_MAX || most < GA_PAIRS_PER_WG || ga_pair_lds_ints(n) * sizeof(int32_t) * (size_t)most > 60 * 1024) {
        // (too many pairs for one round of workgroups -- or tours too long for that many pairs' LDS --, or so few that eight a workgroup already fit:
        // eight per workgroup)
        groups = (pairs + GA_PAIRS_PER_WG - 1) / GA_PAIRS_PER_WG;
        base = GA_PAIRS_PER_WG; extra = 0; per_wg = GA_PAIRS_PER_WG;
        return (unsigned)groups;
    }
    base = pairs / groups; extra = pairs % groups; per_wg = base + (extra ? 1 : 0);
    return (unsigned)groups;
}

bool ga_generation_fits(int n, int pop)
{
    return pop <= GA_LDS_POP && ga_pair_lds_ints(n) * sizeof(int32_t) * GA_PAIRS_PER_WG <= 60 * 1024;
}

int launch_ga_generation(hipStream_t st, int n, int pop, const double *D, const int32_t *cur, const double *cur_fit, const double *cur_dist,
                         int32_t *nxt, double *nxt_fit, double *nxt_dist, const fcpp_ga_config &cfg, int gen, GaState *state,
                         int32_t *best_route, double *hist)
{
    const size_t pl = ga_pair_lds_ints(n);
    int base = 0, extra = 0, per_wg = 0;
    const unsigned blocks = ga_pair_groups(n, pop, base, extra, per_wg) + 2u;
    const size_t lds = std::max(pl * sizeof(int32_t) * (size_t)per_wg, (size_t)pop * sizeof(double));
    hipLaunchKernelGGL(k_ga_generation, dim3(blocks), dim3(SB), lds, st, n, pop, D, cur, cur_fit, cur_dist, nxt, nxt_fit, nxt_dist, cfg, gen,
                       (int)pl, base, extra, state, best_route, hist);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp

#ifdef FCPP_DIAG_GA
extern "C" __attribute__((visibility("default"))) int fcpp_diag_ga_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcpp::g_ga_stamps), sizeof(fcpp::g_ga_stamps));
}
#endif
