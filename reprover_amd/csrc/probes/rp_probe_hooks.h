// PROBE BUILDS ONLY (-DRP_PHASE_PROBE; tools/probes/gemm_phase.py).  The product build never includes this file: in it
// every hook below is an empty statement (rp_gemm.h, rp_encoder_kernels.h).  Results of probe builds are timing only - the
// ablations compute wrong values on purpose.
//
//   RP_TS(slot)                 thread 0 of every workgroup records the 100 MHz wall clock at four points of its tile
//                               (0 start, 1 first k-tile landed, 2 main loop done, 3 epilogue done)
//   RP_HTS(kt, which)           hand-over detail of workgroup 1000, every wave: shader clock before the waits / after the
//                               counted vmcnt wait / after the barrier
//   RP_PROBE_STAGE_FILTER       -DRP_PROBE_NO_DMA: no operand DMA after the first two k-tiles; _HALF_DMA: half the bytes;
//                               _SAME_TILE: every refill re-reads the first two k-tiles (L2 hits)
//   RP_PROBE_READS_*            -DRP_PROBE_NO_READS: no fragment read after the prologue's
//   RP_PTS*, RP_PROBE_PERSIST_TILE_END   persistent form: per-workgroup SUMS over its tiles [prologue, main loop,
//                               epilogue + end barrier, tiles]
//   RP_ABL_*                    epilogue ablations (rp_encoder_kernels.h): -DRP_ABL_NOLOAD / _NOSTORE / _NOGELU
#pragma once

// (included from rp_gemm.h INSIDE namespace rp)
__device__ unsigned long long g_phase_ts[4 * 16384];
__device__ unsigned long long g_handover_ts[8 * 64 * 3];  // [wave][kt][0..2]

#define RP_TS(slot)                                                                                    \
  do {                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_phase_ts[blockIdx.x * 4 + (slot)] = wall_clock64();  \
  } while (0)
// the plain loop (gemm_tile) stamps like the pipelined one unless -DRP_PROBE_NO_PLAIN_TS (the scan's sample pass would
// overwrite the per-workgroup sums of a persistent filter pass)
#ifdef RP_PROBE_NO_PLAIN_TS
#define RP_TS_PLAIN(slot) ((void)0)
#else
#define RP_TS_PLAIN(slot) RP_TS(slot)
#endif
#define RP_HTS(kt, which)                                                              \
  do {                                                                                 \
    if (blockIdx.x == 1000 && (threadIdx.x & 63) == 0 && (kt) < 64)                     \
      g_handover_ts[((threadIdx.x >> 6) * 64 + (kt)) * 3 + (which)] = clock64();       \
  } while (0)

#if defined(RP_PROBE_NO_DMA)
#define RP_PROBE_STAGE_FILTER(kt, half) \
  do {                                  \
    if ((kt) > 1) return;               \
  } while (0)
#elif defined(RP_PROBE_HALF_DMA)
#define RP_PROBE_STAGE_FILTER(kt, half)   \
  do {                                    \
    if ((kt) > 1 && (half) == 1) return;  \
  } while (0)
#elif defined(RP_PROBE_SAME_TILE)
#define RP_PROBE_STAGE_FILTER(kt, half) \
  do {                                  \
    kt = kt & 1;                        \
  } while (0)
#else
#define RP_PROBE_STAGE_FILTER(kt, half) ((void)0)
#endif

#if defined(RP_PROBE_NO_READS)
#define RP_PROBE_READS_DECL bool probe_reads_on = true
#define RP_PROBE_READS_GATE   \
  do {                        \
    if (!probe_reads_on) return; /* keep what the prologue read */ \
  } while (0)
#define RP_PROBE_READS_PRIME(read_frags, smem) \
  do {                                         \
    read_frags(smem, 1, 1);                    \
    probe_reads_on = false;                    \
  } while (0)
#else
#define RP_PROBE_READS_DECL ((void)0)
#define RP_PROBE_READS_GATE ((void)0)
#define RP_PROBE_READS_PRIME(read_frags, smem) ((void)0)
#endif

#define RP_PTS_DECL unsigned long long p_t0 = 0, p_t1 = 0, p_t2 = 0
#define RP_PTS(v) v = wall_clock64()
#define RP_PROBE_PERSIST_TILE_END(more)                       \
  do {                                                        \
    if (more) {                                               \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
      __builtin_amdgcn_s_barrier();                           \
    }                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 4096) {              \
      const unsigned long long p_t3 = wall_clock64();         \
      g_phase_ts[blockIdx.x * 4 + 0] += p_t1 - p_t0;          \
      g_phase_ts[blockIdx.x * 4 + 1] += p_t2 - p_t1;          \
      g_phase_ts[blockIdx.x * 4 + 2] += p_t3 - p_t2;          \
      g_phase_ts[blockIdx.x * 4 + 3] += 1;                    \
    }                                                         \
  } while (0)

// ---- epilogue ablations (rp_encoder_kernels.h) -------------------------------------------------------------------
// -DRP_ABL_NOLOAD: the residual epilogue without the reads of the old planes (values from the address, so nothing folds)
#define RP_ABL_NOLOAD_BODY(xh, xl, off)               \
  do {                                                \
    xh = make_uint4((uint32_t)(off), 0u, 0u, 0u);     \
    xl = {};                                          \
  } while (0)
// -DRP_ABL_NOSTORE: an epilogue without its global stores (the values are kept alive)
#define RP_ABL_KEEP6(a, b, c, d, e, f) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f))
#define RP_ABL_KEEP4(a, b, c, d) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d))
// -DRP_ABL_NOGELU: the gated-GELU epilogue without the activation's arithmetic
#define RP_ABL_NOGELU_BODY(g, u) ((g) * (u))
