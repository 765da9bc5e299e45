// Microbenchmark 12 (round 4): which MFMA shape should the consumer waves of the producer/consumer 3x3 kernel use?
//
// Same structure as conv_m32p_kernel: 512 threads, waves 0-3 consumers (LDS fragment reads + MFMAs only), waves 4-7
// producers (ND LDS-DMA instructions of 1 KiB per wave and stage), one workgroup barrier per 16-channel stage, every
// operand re-read from LDS, random bf16 data.  A stage is 2 planes x 9 taps = 18 (plane, tap) groups of 8 channels for a
// 96-row Cout block and a 384-pixel tile: 81 v_mfma_f32_32x32x16 per consumer wave (MR = 3 x NR = 3 accumulators,
// 6 ds_read_b128 per 9 MFMAs) or 162 v_mfma_f32_16x16x32 (6 x 6 accumulators, 12 ds_read_b128 per 36 MFMAs: the same
// LDS bytes per flop), whose K = 32 is FOUR groups -- any four (plane, tap) pairs, since each 16-lane group of a wave
// reads its own fragment address -- so 18 groups are 4.5 k-steps.
//   SHAPE 0: 32x32x16, 3 x 3, 9 taps per stage (the kernel as shipped in round 3)
//   SHAPE 1: 16x16x32, 6 x 6, stages alternate 5 / 4 k-steps (the odd group pair straddles two stages)
//   SHAPE 2: 16x16x32, 6 x 6, 5 k-steps per stage (two zero groups of padding: +11 % MFMAs)
//   SHAPE 3: 16x16x32, 6 x 5 (320-pixel tile: 11 reads per 30 MFMAs, +10 % LDS bytes per flop, 88 fragment registers)
//   SHAPE 4: SHAPE 1 with the fragment reads of the next k-step spread BETWEEN the MFMAs of a column (at most one read per
//            two MFMAs) instead of issued in a burst behind the column: an in-order wave that issues seven ds_read_b128
//            back to back leaves its matrix pipe idle once the last queued MFMA has drained
// Reported per variant: us per stage, shader cycles per stage, clock, useful PFLOP/s (padding MFMAs not counted).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int WBUF = 18 * 96 * 16;        // one stage of packed weights: 27 648 B
constexpr int XPL = 18 * 26 * 16;         // one 8-channel plane of an 18 x 26 halo tile
constexpr int XBUF = 2 * XPL;
constexpr int OFF_X = 2 * WBUF;
constexpr int LDS_BYTES = OFF_X + 2 * XBUF + 64 * 1024;   // + landing zone of the producers' DMA

template <int OFF> __device__ __forceinline__ void rd(bf16x8& d, uint32_t a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
__device__ __forceinline__ void mf32(f32x16& c, const bf16x8& a, const bf16x8& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mf16(f32x4& c, const bf16x8& a, const bf16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
// address arithmetic the compiler must not hoist out of the k-step that uses it (30 live addresses otherwise)
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c_uniform) {
  uint32_t d;
  asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
  return d;
}
template <int N> __device__ __forceinline__ void landed(bf16x8* a, bf16x8* b, int nb) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
  for (int i = 0; i < N; ++i) if (i < nb) asm volatile("" : "+v"(b[i]));
}

template <int SHAPE, int ND>
__global__ __launch_bounds__(512, 2) void k(const char* src, int iters, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool producer = wave_all >= 4;
  const int wave = wave_all & 3;
  for (int i = tid; i < LDS_BYTES / 16; i += 512) ((uint4*)smem)[i] = ((const uint4*)src)[i];
  __syncthreads();
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  if (producer) {
    __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < iters; ++it) {
      const char* s = src + (size_t)((it * 7 + blockIdx.x) & 63) * (ND > 0 ? 4 * ND * 1024 : 1024);
      char* dst = smem + OFF_X + 2 * XBUF + (it & 1) * 32768;
#pragma unroll
      for (int j = 0; j < ND; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * 4 + wave) * 1024 + lane * 16), (lds_void_t*)(dst + (j * 4 + wave) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if constexpr (SHAPE == 0) {
    const int half = lane >> 5, r = lane & 31;
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    uint32_t xb[3];
    for (int n = 0; n < 3; ++n) {
      const int p = (wave * 3 + n) * 32 + r;
      xb[n] = (uint32_t)(size_t)smem + OFF_X + half * XPL + ((p / 24) * 26 + p % 24) * 16;
    }
    bf16x8 a0[3], b0[3], a1[3], b1[3];
    for (int it = 0; it < iters; ++it) {
      const uint32_t wa = (uint32_t)(size_t)smem + (it & 1) * WBUF + (half * 96 + r) * 16;
      const uint32_t xo = (it & 1) * XBUF;
      rd<0>(a0[0], wa); rd<512>(a0[1], wa); rd<1024>(a0[2], wa);
      rd<0>(b0[0], xb[0] + xo); rd<0>(b0[1], xb[1] + xo); rd<0>(b0[2], xb[2] + xo);
      landed<3>(a0, b0, 3);
#define TAP(T, CA, CB, NA, NB, MORE)                                                             \
      mf32(acc[0], CA[0], CB[0]); mf32(acc[1], CA[1], CB[0]); mf32(acc[2], CA[2], CB[0]);        \
      if (MORE) { rd<(T) * 3072>(NA[0], wa); rd<(T) * 3072 + 512>(NA[1], wa); rd<(T) * 3072 + 1024>(NA[2], wa); \
                  rd<((T) % 3) * 16>(NB[0], xb[0] + xo + ((T) / 3) * 416); rd<((T) % 3) * 16>(NB[1], xb[1] + xo + ((T) / 3) * 416); rd<((T) % 3) * 16>(NB[2], xb[2] + xo + ((T) / 3) * 416); } \
      mf32(acc[3], CA[0], CB[1]); mf32(acc[4], CA[1], CB[1]); mf32(acc[5], CA[2], CB[1]);        \
      mf32(acc[6], CA[0], CB[2]); mf32(acc[7], CA[1], CB[2]); mf32(acc[8], CA[2], CB[2]);        \
      if (MORE) landed<3>(NA, NB, 3);
      TAP(1, a0, b0, a1, b1, 1) TAP(2, a1, b1, a0, b0, 1) TAP(3, a0, b0, a1, b1, 1) TAP(4, a1, b1, a0, b0, 1) TAP(5, a0, b0, a1, b1, 1)
      TAP(6, a1, b1, a0, b0, 1) TAP(7, a0, b0, a1, b1, 1) TAP(8, a1, b1, a0, b0, 1) TAP(0, a0, b0, a1, b1, 0)
#undef TAP
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0;
    for (int i = 0; i < 9; ++i) { asm volatile("" : "+v"(acc[i])); t += acc[i][0] + acc[i][7]; }
    if (t == 123.456f) sink[0] = t;
  } else if constexpr (SHAPE == 4) {
    constexpr int NR = 6, RING = 7;
    const int q = lane >> 4, l15 = lane & 15;
    f32x4 acc[6][NR];
    for (int m = 0; m < 6; ++m) for (int n = 0; n < NR; ++n) for (int j = 0; j < 4; ++j) acc[m][n][j] = 0.f;
    uint32_t xb[NR], koff[5];
    for (int n = 0; n < NR; ++n) {
      const int p = (wave * NR + n) * 16 + l15;
      xb[n] = (uint32_t)(size_t)smem + OFF_X + ((p / 24) * 26 + p % 24) * 16;
    }
    for (int s = 0; s < 5; ++s) {
      const int g = (4 * s + q) % 18, plane = g / 9, tap = g % 9;
      koff[s] = plane * XPL + ((tap / 3) * 26 + tap % 3) * 16;
    }
    bf16x8 a0[6], a1[6], bR[RING];
    // Issue order of a k-step's prefetch (in-order LDS returns): A'0 A'1 B'0 | A'2 A'3 B'1 | A'4 B'2 | A'5 B'3 | B'4 | B'5.
    // Column 0 of the next k-step needs A'0..5 and B'0: all but the three youngest (B'3 B'4 B'5) -> lgkmcnt(3); column 3 needs B'3
    // with B'4 B'5 and the eight requests of the new k-step's columns 0-2 behind it -> lgkmcnt(10); columns 4, 5: lgkmcnt(11).
#define WAITCNT(N) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
#define BREG(S, N) bR[((N) - (S) + 4 * RING) % RING]
#define RDA(S, M, NA) rd<((S) + 1) * 6144 + (M) * 256>(NA[M], wa);
#define WAIT_OF(S, N, MORE) ((S) == 0 ? ((N) == 0 ? 5 : (N) == 1 ? 7 : (N) == 2 ? 9 : (N) == 3 ? 10 : 11) \
                                      : (MORE) ? ((N) == 0 ? 3 : (N) == 3 ? 10 : (N) >= 4 ? 11 : -1)       \
                                               : ((N) == 0 ? 3 : (N) >= 3 ? 5 - (N) : -1))
#define COL(S, N, CA, NA, MORE)                                                                    \
      {                                                                                            \
        if constexpr (WAIT_OF(S, N, MORE) >= 0) { WAITCNT(WAIT_OF(S, N, MORE) < 0 ? 0 : WAIT_OF(S, N, MORE)) } \
        if constexpr ((N) == 0) { _Pragma("unroll") for (int m = 0; m < 6; ++m) asm volatile("" : "+v"(CA[m])); } \
        asm volatile("" : "+v"(BREG(S, N)));                                                       \
        mf16(acc[0][N], CA[0], BREG(S, N)); mf16(acc[1][N], CA[1], BREG(S, N));                    \
        if constexpr (MORE) { if constexpr ((N) == 0) { RDA(S, 0, NA) } else if constexpr ((N) == 1) { RDA(S, 2, NA) }          \
                              else if constexpr ((N) == 2) { RDA(S, 4, NA) } else if constexpr ((N) == 3) { RDA(S, 5, NA) } }   \
        mf16(acc[2][N], CA[2], BREG(S, N)); mf16(acc[3][N], CA[3], BREG(S, N));                    \
        if constexpr (MORE) { if constexpr ((N) == 0) { RDA(S, 1, NA) } else if constexpr ((N) == 1) { RDA(S, 3, NA) } }        \
        mf16(acc[4][N], CA[4], BREG(S, N));                                                        \
        if constexpr (MORE) rd<0>(BREG((S) + 1, N), add3(xb[N], koff[(S) + 1], xo));               \
        mf16(acc[5][N], CA[5], BREG(S, N));                                                        \
      }
#define KSTEP(S, CA, NA, MORE) COL(S, 0, CA, NA, MORE) COL(S, 1, CA, NA, MORE) COL(S, 2, CA, NA, MORE) COL(S, 3, CA, NA, MORE) COL(S, 4, CA, NA, MORE) COL(S, 5, CA, NA, MORE)
#define STAGE_HEAD(PAR)                                                                            \
      const uint32_t wa = (uint32_t)(size_t)smem + (PAR) * WBUF + (q * 96 + l15) * 16;             \
      const uint32_t xo = (PAR) * XBUF;                                                            \
      rd<0>(a0[0], wa); rd<256>(a0[1], wa); rd<512>(a0[2], wa); rd<768>(a0[3], wa); rd<1024>(a0[4], wa); rd<1280>(a0[5], wa); \
      _Pragma("unroll") for (int n = 0; n < NR; ++n) rd<0>(bR[n], add3(xb[n], koff[0], xo));
    for (int it = 0; it < iters; it += 2) {
      {
        STAGE_HEAD(0)
        KSTEP(0, a0, a1, true) KSTEP(1, a1, a0, true) KSTEP(2, a0, a1, true) KSTEP(3, a1, a0, true) KSTEP(4, a0, a1, false)
        __builtin_amdgcn_s_barrier();
      }
      {
        STAGE_HEAD(1)
        KSTEP(0, a0, a1, true) KSTEP(1, a1, a0, true) KSTEP(2, a0, a1, true) KSTEP(3, a1, a0, false)
        __builtin_amdgcn_s_barrier();
      }
    }
#undef KSTEP
#undef COL
#undef STAGE_HEAD
#undef WAITCNT
#undef BREG
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0;
    for (int m = 0; m < 6; ++m) for (int n = 0; n < NR; ++n) { asm volatile("" : "+v"(acc[m][n])); t += acc[m][n][0] + acc[m][n][3]; }
    if (t == 123.456f) sink[0] = t;
  } else {
    // A fragments double-buffered (a0 / a1: all six are live through the six columns of a k-step); B fragments in a ring of
    // NR + 1 registers: column n of k-step s sits in bR[(n - s) mod (NR + 1)], and right after the column's MFMAs the
    // fragment of column n of k-step s + 1 is requested into the register column n - 1 has just released.  LDS returns in
    // order, so counted waits (lgkmcnt(5) in front of column 0, lgkmcnt(NR + 5) in front of the others) expose no latency.
    constexpr int NR = SHAPE == 3 ? 5 : 6;
    constexpr int RING = NR + 1;
    constexpr int TW = SHAPE == 3 ? 20 : 24;     // 16 x 20 or 16 x 24 pixel tile, halo rows of 26 slots either way
    const int q = lane >> 4, l15 = lane & 15;
    f32x4 acc[6][NR];
    for (int m = 0; m < 6; ++m) for (int n = 0; n < NR; ++n) for (int j = 0; j < 4; ++j) acc[m][n][j] = 0.f;
    uint32_t xb[NR], koff[5];
    for (int n = 0; n < NR; ++n) {
      const int p = (wave * NR + n) * 16 + l15;
      xb[n] = (uint32_t)(size_t)smem + OFF_X + ((p / TW) * 26 + p % TW) * 16;
    }
    for (int s = 0; s < 5; ++s) {   // group g = 4 s + q of a stage -> (plane, tap); groups 18, 19 are the padding / straddle pair
      const int g = (4 * s + q) % 18, plane = g / 9, tap = g % 9;
      koff[s] = plane * XPL + ((tap / 3) * 26 + tap % 3) * 16;
    }
    bf16x8 a0[6], a1[6], bR[RING];
#define ISSUE_A(S, NA)                                                                             \
      rd<(S) * 6144>(NA[0], wa); rd<(S) * 6144 + 256>(NA[1], wa); rd<(S) * 6144 + 512>(NA[2], wa); \
      rd<(S) * 6144 + 768>(NA[3], wa); rd<(S) * 6144 + 1024>(NA[4], wa); rd<(S) * 6144 + 1280>(NA[5], wa);
#define WAITCNT(N) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
#define BREG(S, N) bR[((N) - (S) + 4 * RING) % RING]
#define COL(S, N, CA, NA, MORE)                                                                    \
      if constexpr ((N) < NR) {                                                                    \
        WAITCNT((N) == 0 ? NR - 1 : ((MORE) ? NR + 5 : NR - 1 - (N)))                              \
        if constexpr ((N) == 0) { _Pragma("unroll") for (int m = 0; m < 6; ++m) asm volatile("" : "+v"(CA[m])); } \
        asm volatile("" : "+v"(BREG(S, N)));                                                       \
        _Pragma("unroll") for (int m = 0; m < 6; ++m) mf16(acc[m][N], CA[m], BREG(S, N));          \
        if constexpr (MORE) {                                                                      \
          if constexpr ((N) == 0) { ISSUE_A((S) + 1, NA) }                                         \
          rd<0>(BREG((S) + 1, N), add3(xb[N], koff[(S) + 1], xo));                                 \
        }                                                                                          \
      }
#define KSTEP(S, CA, NA, MORE) COL(S, 0, CA, NA, MORE) COL(S, 1, CA, NA, MORE) COL(S, 2, CA, NA, MORE) COL(S, 3, CA, NA, MORE) COL(S, 4, CA, NA, MORE) COL(S, 5, CA, NA, MORE)
#define STAGE_HEAD(PAR)                                                                            \
      const uint32_t wa = (uint32_t)(size_t)smem + (PAR) * WBUF + (q * 96 + l15) * 16;             \
      const uint32_t xo = (PAR) * XBUF;                                                            \
      ISSUE_A(0, a0)                                                                               \
      _Pragma("unroll") for (int n = 0; n < NR; ++n) rd<0>(bR[n], add3(xb[n], koff[0], xo));
    for (int it = 0; it < iters; it += 2) {   // two stages per trip, straight-line: the ring position of every fragment is a compile-time constant
      {
        STAGE_HEAD(0)
        KSTEP(0, a0, a1, true) KSTEP(1, a1, a0, true) KSTEP(2, a0, a1, true) KSTEP(3, a1, a0, true) KSTEP(4, a0, a1, false)
        __builtin_amdgcn_s_barrier();
      }
      {
        STAGE_HEAD(1)
        if constexpr (SHAPE == 2) {
          KSTEP(0, a0, a1, true) KSTEP(1, a1, a0, true) KSTEP(2, a0, a1, true) KSTEP(3, a1, a0, true) KSTEP(4, a0, a1, false)
        } else {
          KSTEP(0, a0, a1, true) KSTEP(1, a1, a0, true) KSTEP(2, a0, a1, true) KSTEP(3, a1, a0, false)
        }
        __builtin_amdgcn_s_barrier();
      }
    }
#undef KSTEP
#undef COL
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0;
    for (int m = 0; m < 6; ++m) for (int n = 0; n < NR; ++n) { asm volatile("" : "+v"(acc[m][n])); t += acc[m][n][0] + acc[m][n][3]; }
    if (t == 123.456f) sink[0] = t;
  }
  if (blockIdx.x == 0 && lane == 0 && wave_all == 0) cyc[0] = __builtin_amdgcn_s_memtime() - t_begin;
}

template <int SHAPE, int ND>
double run(const char* name, const char* d, float* sink, unsigned long long* cyc, bool print = true) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<SHAPE, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<SHAPE, ND><<<256, 512, LDS_BYTES>>>(d, 200, sink, cyc);
  hipEventRecord(a);
  k<SHAPE, ND><<<256, 512, LDS_BYTES>>>(d, iters, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / iters;
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double useful = (SHAPE == 3 ? 4.5 * 30 : 162.0) * 16384.0 * 4 * 256;   // flops per stage, chip-wide
  if (print)
    printf("%-44s ND=%2d  %6.3f us/stage  %6.0f cyc/stage  clk %.2f GHz  %5.3f PFLOP/s useful\n", name, ND, us, (double)c / iters,
           (double)c / iters / (us * 1e3), useful / (us * 1e-6) / 1e15);
  hipEventDestroy(a); hipEventDestroy(b);
  return us;
}

int main() {
  const size_t nbytes = 4 << 20;
  char* d; hipMalloc(&d, nbytes);
  unsigned short* h = (unsigned short*)malloc(nbytes);
  srand(12345);
  for (size_t i = 0; i < nbytes / 2; ++i) {   // N(0, 1) rounded to bf16: full-range mantissas and signs (zeros / constants overstate the clock)
    const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    const float f = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
  }
  hipMemcpy(d, h, nbytes, hipMemcpyHostToDevice);
  float* sink; hipMalloc(&sink, 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  // warm the chip up (clock / power state), then interleave the variants three times (rule: same process, same box)
  for (int w = 0; w < 3; ++w) run<0, 7>("warm-up", d, sink, cyc, false);
  for (int round = 0; round < 3; ++round) {
    printf("-- round %d\n", round);
    run<0, 0>("32x32x16 3x3, 81 MFMA/stage", d, sink, cyc);
    run<1, 0>("16x16x32 6x6, 5/4 k-steps", d, sink, cyc);
    run<2, 0>("16x16x32 6x6, 5 k-steps (padded)", d, sink, cyc);
    run<3, 0>("16x16x32 6x5, 5/4 k-steps", d, sink, cyc);
    run<4, 0>("16x16x32 6x6, 5/4 k-steps, reads spread", d, sink, cyc);
    run<0, 7>("32x32x16 3x3, 81 MFMA/stage", d, sink, cyc);
    run<1, 7>("16x16x32 6x6, 5/4 k-steps", d, sink, cyc);
    run<2, 7>("16x16x32 6x6, 5 k-steps (padded)", d, sink, cyc);
    run<3, 7>("16x16x32 6x5, 5/4 k-steps", d, sink, cyc);
    run<4, 7>("16x16x32 6x6, 5/4 k-steps, reads spread", d, sink, cyc);
    run<0, 14>("32x32x16 3x3, 81 MFMA/stage", d, sink, cyc);
    run<1, 14>("16x16x32 6x6, 5/4 k-steps", d, sink, cyc);
    run<2, 14>("16x16x32 6x6, 5 k-steps (padded)", d, sink, cyc);
    run<3, 14>("16x16x32 6x5, 5/4 k-steps", d, sink, cyc);
    run<4, 14>("16x16x32 6x6, 5/4 k-steps, reads spread", d, sink, cyc);
  }
  return 0;
}
