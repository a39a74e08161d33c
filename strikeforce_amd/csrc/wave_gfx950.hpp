// wave_gfx950.hpp — the CDNA4 (gfx950) wavefront backend of sf_core.hpp.
//
// One workgroup = one 64-lane wavefront = one arena.  Per-lane values are plain registers; cross-lane
// traffic uses v_readlane / DPP / s_ballot, never LDS.  Wave-uniform values produced here (ballots,
// readlane, readfirstlane) are SGPRs, so the slot-ordered loops of the core compile to scalar control
// flow with the vector unit doing the per-entity predicated updates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SF_DEV __device__ __forceinline__

#if (defined(SF_EXP_SHORT_ROUND) || defined(SF_EXP_FREE_WARMUP) || defined(SF_EXP_ZLOOP_FREE)) && !defined(SF_EXPERIMENT_BUILD)
#error "timing experiments (wrong results by construction): only together with -DSF_EXPERIMENT_BUILD"
#endif

namespace sf {

// HBM pointers reach the kernels inside a by-value struct, where clang leaves them in the generic address
// space (flat_load/flat_store, which also tie up the LDS counter).  Every HBM access below goes through these
// casts so that it is a global_load/global_store.
#define SF_GLOBAL __attribute__((address_space(1)))
template <class T>
static __device__ __forceinline__ const SF_GLOBAL T *gptr(const T *p) {
  return (const SF_GLOBAL T *)p;
}
template <class T>
static __device__ __forceinline__ SF_GLOBAL T *gptr(T *p) {
  return (SF_GLOBAL T *)p;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // 16 B per lane; a builtin vector, usable in any address space

#define SF_COUNT(k)  // event counters of the CPU emulator (tests/emu/wave_emu.hpp): nothing on the device
#define SF_PROF(ph)  // phase markers of the CPU emulator's op profile (tests/emu/wave_emu.hpp): nothing on the device

// In-kernel phase stamps: compiled in only by the diagnostic build of tools/diag_stamps.sh (-DSF_DIAG_STAMPS), which
// is never the product library.  Wave cycles (s_memtime) between consecutive stamps are charged to the phase named at
// the later stamp, in lane `phase` of one register; the totals go to a buffer of their own (sf_diag_buffer).
#ifdef SF_DIAG_STAMPS
extern __device__ uint32_t sf_diag_buffer[];
#define SF_STAMP_BEGIN(S) ((S).dacc = 0u, (S).dlast = (uint32_t)__builtin_amdgcn_s_memtime())
#define SF_STAMP(S, ph)                                                              \
  do {                                                                               \
    const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime();                      \
    (S).dacc = threadIdx.x == (ph) ? (S).dacc + (t_ - (S).dlast) : (S).dacc;         \
    (S).dlast = t_;                                                                  \
  } while (0)
#define SF_STAMP_END(S, a) \
  do {                     \
    if (threadIdx.x < 16u) ::sf::sf_diag_buffer[(size_t)(a) * 16u + threadIdx.x] = (S).dacc; \
  } while (0)
// absolute times (100 MHz) of a k_step wave: state loaded, steps done (sf_api.hip sf_diag_times; tools/r04_k1_times.py)
extern __device__ unsigned long long sf_diag_times[];
#define SF_STAMP_LOADED()                                                                    \
  do {                                                                                       \
    __builtin_amdgcn_s_waitcnt(0);                                                           \
    if (threadIdx.x == 0) ::sf::sf_diag_times[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define SF_STAMP_STEPPED()                                                                   \
  do {                                                                                       \
    if (threadIdx.x == 0) ::sf::sf_diag_times[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define SF_STAMP_BEGIN(S)
#define SF_STAMP(S, ph)
#define SF_STAMP_END(S, a)
#define SF_STAMP_LOADED()
#define SF_STAMP_STEPPED()
#endif

#ifndef SF_RNG_PRIO
#define SF_RNG_PRIO 1
#endif
#ifndef SF_RNG_PRIO_LEVEL
#define SF_RNG_PRIO_LEVEL 3
#endif
#define SF_STR2(x) #x
#define SF_STR(x) SF_STR2(x)

struct WaveGfx950 {
  using V = uint32_t;
  using P = bool;

  static SF_DEV V lane() { return threadIdx.x; }  // one 64-lane wavefront per workgroup
  static SF_DEV uint64_t ballot(P p) { return __builtin_amdgcn_ballot_w64(p); }
  static SF_DEV int ctz64(uint64_t m) { return __builtin_ctzll(m); }
  static SF_DEV int clz64(uint64_t m) { return __builtin_clzll(m); }
  static SF_DEV uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
  static SF_DEV uint32_t readlane(const V &v, uint32_t idx) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)uni(idx));
  }
  static SF_DEV void setlane(V &v, uint32_t idx, uint32_t val) {
    v = (lane() == uni(idx)) ? val : v;  // v_cmp + v_cndmask (this clang has no writelane builtin)
  }
  static SF_DEV V select(P p, V a, V b) { return p ? a : b; }
  static SF_DEV V sar31(V v) { return (uint32_t)((int32_t)v >> 31); }
  static SF_DEV P le0(V v) { return (int32_t)v <= 0; }
  static SF_DEV P ltu(V a, V b) { return a < b; }
  static SF_DEV P frombits(uint64_t m) { return (m >> lane()) & 1ull; }
  static SF_DEV P all() { return true; }
  // sum over lanes 0..17 of a value that is zero on lanes >= 18, left on the vector unit: valid on lanes 16..31
  // (DPP reductions inside each 16-lane row, then row 1 += row 0's total via row_bcast:15)
  static SF_DEV V sum18_row1(V v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0x2, 0xf, false);  // row_bcast:15 into row 1 only
    return (uint32_t)x;
  }
  static SF_DEV V minu(V a, V b) { return a < b ? a : b; }
  static SF_DEV V shrv(V a, V sh) { return a >> sh; }  // per-lane shift amounts (< 32)
  static SF_DEV V shlv(V a, V sh) { return a << sh; }
  static SF_DEV P gts(V a, V b) { return (int32_t)a > (int32_t)b; }  // signed
  static SF_DEV int popc64(uint64_t m) { return __builtin_popcountll(m); }
  // cell bitmaps in LDS: OR bits into words (lanes may share a word: LDS atomics), with or without the old value
  static SF_DEV void lds_or_u32(uint32_t *base, V widx, V bits, P pred) {
    if (pred) (void)__hip_atomic_fetch_or(base + widx, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
  }
  static SF_DEV V lds_or_rtn_u32(uint32_t *base, V widx, V bits, P pred) {
    uint32_t old = 0u;
    if (pred) old = __hip_atomic_fetch_or(base + widx, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
    return old;
  }
  static SF_DEV void lds_store_u32(uint32_t *base, V widx, V val, P pred) {
    if (pred) base[widx] = val;
    __builtin_amdgcn_wave_barrier();
  }
  static SF_DEV void lds_zero(uint32_t *base, uint32_t nwords) {  // nwords: multiple of 4
    for (uint32_t off = lane() * 4u; off < nwords; off += 64u * 4u)
      *reinterpret_cast<u32x4 *>(base + off) = (u32x4)(0u);
    __builtin_amdgcn_wave_barrier();
  }
  static SF_DEV V mad24(V a, uint32_t b, V c) { return __umul24(a, b) + c; }
  // per-lane byte store into the LDS flag plane (distinct cells per lane)
  static SF_DEV void lds_store_u8(uint8_t *lds, V idx, V val, P pred) {
    if (pred) lds[idx] = (uint8_t)val;
    __builtin_amdgcn_wave_barrier();
  }
  // low 32 bits of a product whose operands are below 2^24 (v_mul_u32_u24)
  static SF_DEV V mul24(V a, V b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // (the compiler picks the 32-bit multiply for __umul24)
    return r;
  }
  // lane i <- lane i + 1 (wave_shl:1, a gfx9 DPP control); lane 63 reads 0
  static SF_DEV V shl1(V v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }

  // LDS flag plane
  static SF_DEV V lds_u8(const uint8_t *lds, V idx, P pred) { return pred ? (uint32_t)lds[idx] : 0u; }
  static SF_DEV V lds_u8_any(const uint8_t *lds, V idx) { return (uint32_t)lds[idx]; }  // every lane: idx must be valid
  static SF_DEV uint32_t ulds_u8(const uint8_t *lds, uint32_t idx) { return uni((uint32_t)lds[idx]); }
  static SF_DEV V lds_u32(const uint32_t *lds, V idx, P pred) { return pred ? lds[idx] : 0u; }
  // wave-uniform dword in LDS (the zombie tables of worlds with more than 64 zombies, sf_core.hpp ZL)
  static SF_DEV uint32_t ulds_u32(const uint32_t *lds, uint32_t idx) { return uni(lds[idx]); }
  static SF_DEV void ulds_store_u32(uint32_t *lds, uint32_t idx, uint32_t val) {
    lds[idx] = val;  // every lane writes the same dword
    __builtin_amdgcn_wave_barrier();
  }
  // the same with a wave-uniform first factor, which stays in an SGPR (VOP2 src0)
  static SF_DEV V mul24_su(uint32_t a, V b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
  }
  // --- the generator's hot path, hand-written where the compiler's version carried extra instructions (sf_core.hpp draw()) ---
  // rl <- (lane 17 or 18) ? e * la : rl of the lane above (wave_shl:1): one multiply and one DPP select.  Only the low
  // 16 bits of the product matter and they depend on the low 16 bits of e alone, so e needs no mask.
  // SF_RNG_PRIO: 0 = off, 1 = the wave runs at raised priority from the round's first instruction to its last vector
  // instruction, 2 = until the table load has been issued (rng_prio_end)
  static SF_DEV void rng_prio_end() {
#if SF_RNG_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
  }
  static SF_DEV V rng_commit(V rl, uint32_t e, V la) {
    uint32_t t;
    // (volatile: the block also raises the wave's priority; its partner in rng_reduce lowers it again, and neither may be
    // dropped, duplicated or moved across the other)
    asm volatile(
#if SF_RNG_PRIO
        "s_setprio " SF_STR(SF_RNG_PRIO_LEVEL) "\n\t"
#endif
        "v_mul_u32_u24 %[t], %[e], %[la]\n\t"
        "s_mov_b64 vcc, 0x60000\n\t"
        "v_cndmask_b32_dpp %[rl], %[rl], %[t], vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : [rl] "+v"(rl), [t] "=&v"(t)
        : [e] "s"(e), [la] "v"(la)
        : "vcc");
    return rl;
  }
  // x = d * us + bias on every lane, summed like sum18_row1 (row 1 = rows 0 + 1), reduced mod 65537 to
  // t = lo16 - hi16; returns 2 t + bias, a non-negative byte offset (the caller's table pointer is moved down by
  // the same constant: the load's 32-bit register offset is unsigned and t can be negative).  One mad, five DPP adds
  // (each needs two wait states behind the write it reads: s_nop 1, slots the other waves of the SIMD fill), one SDWA
  // subtract, one shift-add.  `bias` is a vector register holding the same even constant on every lane (a VOP3
  // instruction takes no literal on gfx9, and scalar registers are the scarcer kind in this kernel).
  static SF_DEV V rng_reduce(V d, V us, V bias) {
    uint32_t x;
    asm volatile("v_mad_i32_i24 %[x], %[d], %[us], %[bias]\n\t"
        "s_nop 1\n\t"
#ifndef SF_EXP_SHORT_ROUND  // (timing experiment, round 4: what four vector instructions less per round would buy; wrong sums)
        "v_add_u32_dpp %[x], %[x], %[x] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %[x], %[x], %[x] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %[x], %[x], %[x] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %[x], %[x], %[x] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
#endif
        "v_add_u32_dpp %[x], %[x], %[x] row_bcast:15 row_mask:0x2 bank_mask:0xf\n\t"
        "v_sub_u32_sdwa %[x], %[x], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_lshl_add_u32 %[x], %[x], 1, %[bias]"
#if SF_RNG_PRIO == 1
        "\n\ts_setprio 0"
#endif
        : [x] "=&v"(x)
        : [d] "v"(d), [us] "v"(us), [bias] "v"(bias));
    return x;
  }
  // The whole round of draw() in two blocks around its pair of LDS lookups (round 3).  Written as separate small asm
  // statements the compiler put a hazard `s_nop` behind each of them (it cannot look into inline asm) and the value of
  // the draw was extracted behind the round; here the statements are one block each, and the extraction — v_readlane of
  // lane 18's power, sign fix, & 1023 — sits in the wait states that two of the DPP adds need anyway: 29 instead of 34
  // instructions per draw, same arithmetic, same results (the GPU parity suite compares the generator's registers after
  // every step).  Returns the byte offset of the log lookup; `out` = the draw's value.
  static constexpr bool FUSED_ROUND = true;
  static SF_DEV V rng_round(V &rl, uint32_t e, V la, V seed, V us, V bias, const uint32_t *xt, uint32_t &out) {
    typedef const __attribute__((address_space(3))) uint32_t *lptr;
    const uint32_t base = (uint32_t)(uintptr_t)(lptr)xt;
    uint32_t t, m, oa, ob;
    asm volatile(
#if SF_RNG_PRIO
        "s_setprio " SF_STR(SF_RNG_PRIO_LEVEL) "\n\t"
#endif
        "v_mul_u32_u24 %[t], %[e], %[la]\n\t"
        "s_mov_b64 vcc, 0x60000\n\t"
        "v_cndmask_b32_dpp %[rl], %[rl], %[t], vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_u32_u24 %[m], %[rl], %[seed]\n\t"
        "v_lshlrev_b32_sdwa %[oa], 2, %[m] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %[ob], 2, %[m] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
        : [rl] "+v"(rl), [t] "=&v"(t), [m] "=&v"(m), [oa] "=&v"(oa), [ob] "=&v"(ob)
        : [e] "s"(e), [la] "v"(la), [seed] "v"(seed)
        : "vcc");
    const uint32_t a = *(lptr)(uintptr_t)(oa | base);
    const uint32_t b = *((lptr)(uintptr_t)(ob | base) + 256);  // second half: DS offset 1024
    uint32_t pr, d, x, o, sg;
    asm volatile(
        "v_mul_u32_u24 %[pr], %[a], %[b]\n\t"
        "v_sub_u32_sdwa %[d], %[pr], %[pr] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_mad_i32_i24 %[x], %[d], %[us], %[bias]\n\t"
        "v_readlane_b32 %[o], %[d], 18\n\t"   // lane 18 carries seed 1 and the newest log: its power is the draw's value
        "s_lshr_b32 %[sg], %[o], 31\n\t"      // (signed residue lo16 - hi16: +65537 if negative; & 1023 keeps +1 of it)
#ifndef SF_EXP_SHORT_ROUND
        "v_add_u32_dpp %[x], %[x], %[x] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#endif
        "s_add_i32 %[o], %[o], %[sg]\n\t"
        "s_and_b32 %[o], %[o], 0x3ff\n\t"
#ifndef SF_EXP_SHORT_ROUND
        "v_add_u32_dpp %[x], %[x], %[x] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %[x], %[x], %[x] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %[x], %[x], %[x] row_mirror row_mask:0xf bank_mask:0xf\n\t"
#endif
        "s_nop 1\n\t"
        "v_add_u32_dpp %[x], %[x], %[x] row_bcast:15 row_mask:0x2 bank_mask:0xf\n\t"
        "v_sub_u32_sdwa %[x], %[x], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_lshl_add_u32 %[x], %[x], 1, %[bias]"
#if SF_RNG_PRIO
        "\n\ts_setprio 0"
#endif
        : [pr] "=&v"(pr), [d] "=&v"(d), [x] "=&v"(x), [o] "=&s"(o), [sg] "=&s"(sg)
        : [a] "v"(a), [b] "v"(b), [us] "v"(us), [bias] "v"(bias)
        : "scc");
    out = o;
    return x;
  }
  // 3^lo * 3^(256 hi) from the two halves of the power table (1 KiB each, the table 2 KiB-aligned in LDS), for an
  // exponent pre-scaled by 4: the byte offsets are bit fields of m4 OR-ed into the table's address (v_and_or_b32)
  // the same for an unscaled exponent m (bits above 16 ignored): the two table offsets 4 * byte0(m) and 4 * byte1(m) are
  // one SDWA shift each (a byte select on the operand), against three instructions for the mask / shift / mask form
  static SF_DEV V pow_bytes(const uint32_t *xt, V m) {
    typedef const __attribute__((address_space(3))) uint32_t *lptr;
    const uint32_t base = (uint32_t)(uintptr_t)(lptr)xt;
    uint32_t oa, ob;
    asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(oa) : "v"(m));
    asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ob) : "v"(m));
    const uint32_t a = *(lptr)(uintptr_t)(oa | base);
    const uint32_t b = *((lptr)(uintptr_t)(ob | base) + 256);  // second half: DS offset 1024
    return mul24(a, b);
  }
  static SF_DEV V pow_pair(const uint32_t *xt, V m4) {
    typedef const __attribute__((address_space(3))) uint32_t *lptr;
    const uint32_t base = (uint32_t)(uintptr_t)(lptr)xt;
    const uint32_t a = *(lptr)(uintptr_t)((m4 & 0x3fcu) | base);
    const uint32_t b = *((lptr)(uintptr_t)(((m4 >> 8) & 0x3fcu) | base) + 256);  // second half: DS offset 1024
    return mul24(a, b);
  }
  // 16-bit global load at a 32-bit byte offset from a wave-uniform base (saddr + voffset form), every lane
  static SF_DEV V gload_u16_at(const uint16_t *base, V byte_off) {
    return (uint32_t) * reinterpret_cast<const SF_GLOBAL uint16_t *>(reinterpret_cast<const SF_GLOBAL char *>(gptr(base)) + byte_off);
  }
  static SF_DEV void ulds_store_u8(uint8_t *lds, uint32_t idx, uint32_t val) {
    lds[idx] = (uint8_t)val;  // every lane writes the same byte: no divergence, one LDS pass
    __builtin_amdgcn_wave_barrier();
  }

  // wave-uniform access to the sparse per-cell side tables in HBM (rare path).  Relaxed atomics keep
  // these on the vector memory path, which is coherent with this wave's own earlier stores.
  static SF_DEV int32_t uload_i32(const int32_t *p) {
    return (int32_t)uni((uint32_t)__hip_atomic_load(gptr(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
  }
  static SF_DEV void ustore_i32(int32_t *p, int32_t v) {
    if (lane() == 0u) __hip_atomic_store(gptr(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  static SF_DEV V gload_u16(const uint16_t *base, V idx, P pred) { return pred ? (uint32_t)gptr(base)[idx] : 0u; }
  static SF_DEV int32_t uload_i16(const int16_t *p) {
    return (int32_t)(int16_t)uni((uint32_t)(uint16_t)__hip_atomic_load(gptr(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
  }
  static SF_DEV void ustore_i16(int16_t *p, int16_t v) {
    if (lane() == 0u) __hip_atomic_store(gptr(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }

  // per-lane HBM access (struct-of-arrays: consecutive lanes hit consecutive dwords)
  static SF_DEV V gload(const uint32_t *base, V idx, P pred) { return pred ? gptr(base)[idx] : 0u; }
  static SF_DEV V gload_u8(const uint8_t *base, V idx, P pred) { return pred ? (uint32_t)gptr(base)[idx] : 0u; }
  static SF_DEV void gstore(uint32_t *base, V idx, V val, P pred) {
    if (pred) gptr(base)[idx] = val;
  }

  // flag plane <-> LDS, 16 B per lane per pass (nbytes is a multiple of 16)
  // (four loads in flight before the first LDS store: written one by one, each copy waited for its own load —
  // a 4 KB plane was four round trips in a row)
  static SF_DEV void copy_g2l(uint8_t *lds, const uint8_t *g, uint32_t nbytes) {
    for (uint32_t base = 0u; base < nbytes; base += 4u * 1024u) {
      G2L<4> q;
      const uint32_t n = nbytes - base < 4u * 1024u ? nbytes - base : 4u * 1024u;
      g2l_issue(q, g + base, n);
      g2l_store(q, lds + base, n);
    }
  }
  // the same in two halves, for the caller that has other loads to issue in between: up to U KB, U loads per lane
  template <int U>
  struct G2L {
    u32x4 r[U];
  };
  template <int U>
  static SF_DEV void g2l_issue(G2L<U> &q, const uint8_t *g, uint32_t nbytes) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const uint32_t off = (uint32_t)i * 1024u + lane() * 16u;
      q.r[i] = off < nbytes ? *reinterpret_cast<const SF_GLOBAL u32x4 *>(gptr(g) + off) : (u32x4)(0u);
    }
  }
  template <int U>
  static SF_DEV void g2l_store(const G2L<U> &q, uint8_t *lds, uint32_t nbytes) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const uint32_t off = (uint32_t)i * 1024u + lane() * 16u;
      if (off < nbytes) *reinterpret_cast<u32x4 *>(lds + off) = q.r[i];
    }
    __builtin_amdgcn_wave_barrier();
  }
  static SF_DEV void copy_l2g(uint8_t *g, const uint8_t *lds, uint32_t nbytes) {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t off = lane() * 16u; off < nbytes; off += 64u * 16u)
      *reinterpret_cast<SF_GLOBAL u32x4 *>(gptr(g) + off) = *reinterpret_cast<const u32x4 *>(lds + off);
  }
};

}  // namespace sf
