// tools/microbench/symbol_loop.hip -- cycles per trip of the AC-refinement symbol loop of progressive_stream_kernel
// (w_ac_refine_v4 in jpeglibrary_amd/csrc/k2p_progressive.hip), alone on one wave with synthetic window entries: every entry a plain
// symbol with run 0, so a loop entry makes 63 - zq0 trips before the zero rank runs off the table.  Timed with s_memtime at two
// trip counts; printed: the difference per trip.  Variants:
//   full     the loop as shipped (rotated commits fill the hazard gaps)
//   chain    only what the next trip needs (cur -> entry -> zero rank -> ntab -> cur) and the exit test: the floor
//   nofill   the chain with the four hazard gaps as s_nop
//   hipcc --offload-arch=gfx950 -O3 -o symbol_loop symbol_loop.hip && ./symbol_loop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint64_t tick() {
    uint64_t t;
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int kVariant>
__device__ __forceinline__ uint32_t run(uint32_t lane, uint32_t ent2, uint32_t ntab, uint32_t zq0, uint32_t se, uint32_t &trips) {
    // (every value goes through an empty asm: hipcc lets a tied "+v" operand and an input operand that hold the same known value
    // share ONE register, which the loop then overwrites under the input's feet)
    auto hide = [](uint32_t x) {
        asm volatile("" : "+v"(x));
        return x;
    };
    uint32_t cur = hide(0), zq = zq0, symbits = hide(0), kprev = hide(0), bits = hide(0), cv = hide(0), stop;
    uint32_t scur, st, sn, se_ = 0, rr, t = hide(0), tm, adv, symn = hide(0), u, curn, pl, sg = hide(0), stopr = hide(0);
    uint64_t sok = 0, spb = 0, spp;
    const uint32_t base = hide(0);
    trips = hide(trips);
    if (kVariant == 0) {
        asm volatile(
            "v_mov_b32_e32 %[stop], %[none]\n\t"
            "1:\n\t"
            "v_add_u32_e32 %[trips], 1, %[trips]\n\t"
            "v_readfirstlane_b32 %[scur], %[cur]\n\t"
            "v_and_or_b32 %[pl], %[se_], %[zrl], %[stop]\n\t"
            "v_cndmask_b32_e64 %[symbits], %[symbits], %[symn], %[sok]\n\t"
            "v_cmp_eq_u32_e64 %[spp], %[lane], %[pl]\n\t"
            "v_cndmask_b32_e64 %[bits], %[bits], %[symbits], %[spb]\n\t"
            "v_readlane_b32 %[se_], %[ent2], %[scur]\n\t"
            "v_cndmask_b32_e64 %[kprev], %[kprev], %[stopr], %[sok]\n\t"
            "v_cndmask_b32_e64 %[zq], %[zq], %[t], %[sok]\n\t"
            "v_bfe_u32 %[rr], %[se_], 6, 7\n\t"
            "v_cndmask_b32_e64 %[cv], %[cv], %[sg], %[spp]\n\t"
            "v_add_u32_e32 %[t], %[zq], %[rr]\n\t"
            "v_min_u32_e32 %[tm], 63, %[t]\n\t"
            "v_and_b32_e64 %[adv], 63, %[se_]\n\t"
            "v_readfirstlane_b32 %[st], %[tm]\n\t"
            "v_add_u32_e32 %[symn], %[symbits], %[adv]\n\t"
            "v_bfe_u32 %[sg], %[se_], 14, 16\n\t"
            "v_cmp_gt_u32_e64 %[spb], %[lane], %[kprev]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[sn], %[ntab], %[st]\n\t"
            "s_nop 1\n\t"
            "v_add3_u32 %[curn], %[base], %[symn], %[sn]\n\t"
            "v_or3_b32 %[u], %[t], %[cur], %[sn]\n\t"
            "v_cmp_gt_u32_e64 %[sok], 64, %[u]\n\t"
            "v_add_u32_e32 %[stopr], %[sn], %[t]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[cur], %[cur], %[curn], %[sok]\n\t"
            "v_cndmask_b32_e64 %[stop], %[none], %[stopr], %[sok]\n\t"
            "v_cmp_gt_u32_e32 vcc, %[se], %[stop]\n\t"
            "s_cbranch_vccnz 1b\n\t"
            : [cur] "+v"(cur), [zq] "+v"(zq), [symbits] "+v"(symbits), [kprev] "+v"(kprev), [bits] "+v"(bits), [cv] "+v"(cv),
              [stop] "=&v"(stop), [se_] "+s"(se_), [sok] "+s"(sok), [spb] "+s"(spb), [t] "+v"(t), [symn] "+v"(symn), [sg] "+v"(sg),
              [stopr] "+v"(stopr), [scur] "=&s"(scur), [st] "=&s"(st), [sn] "=&s"(sn), [spp] "=&s"(spp), [rr] "=&v"(rr), [tm] "=&v"(tm),
              [adv] "=&v"(adv), [u] "=&v"(u), [curn] "=&v"(curn), [pl] "=&v"(pl)
            , [trips] "+v"(trips)
            : [ent2] "v"(ent2), [ntab] "v"(ntab), [lane] "v"(lane), [base] "v"(base), [none] "v"(0xFFFFu), [zrl] "v"(0x2000u), [se] "s"(se)
            : "vcc", "memory");
    } else if (kVariant == 1) {
        // the chain and nothing else; gaps left to the hardware's own interlocks where there are none required by the ISA
        // manual they are kept as s_nop (lane select 4, SGPR read 2)
        asm volatile(
            "v_mov_b32_e32 %[stop], %[none]\n\t"
            "1:\n\t"
            "v_add_u32_e32 %[trips], 1, %[trips]\n\t"
            "v_readfirstlane_b32 %[scur], %[cur]\n\t"
            "s_nop 3\n\t"
            "v_readlane_b32 %[se_], %[ent2], %[scur]\n\t"
            "s_nop 1\n\t"
            "v_bfe_u32 %[rr], %[se_], 6, 7\n\t"
            "v_add_u32_e32 %[t], %[zq], %[rr]\n\t"
            "v_min_u32_e32 %[tm], 63, %[t]\n\t"
            "s_nop 0\n\t"
            "v_readfirstlane_b32 %[st], %[tm]\n\t"
            "s_nop 3\n\t"
            "v_readlane_b32 %[sn], %[ntab], %[st]\n\t"
            "s_nop 1\n\t"
            "v_add3_u32 %[curn], %[base], %[symbits], %[sn]\n\t"
            "v_or3_b32 %[u], %[t], %[cur], %[sn]\n\t"
            "v_cmp_gt_u32_e64 %[sok], 64, %[u]\n\t"
            "v_add_u32_e32 %[stopr], %[sn], %[t]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[cur], %[cur], %[curn], %[sok]\n\t"
            "v_cndmask_b32_e64 %[zq], %[zq], %[t], %[sok]\n\t"
            "v_cndmask_b32_e64 %[stop], %[none], %[stopr], %[sok]\n\t"
            "v_cmp_gt_u32_e32 vcc, %[se], %[stop]\n\t"
            "s_cbranch_vccnz 1b\n\t"
            : [cur] "+v"(cur), [zq] "+v"(zq), [symbits] "+v"(symbits), [stop] "=&v"(stop), [se_] "+s"(se_), [sok] "+s"(sok), [t] "+v"(t),
              [stopr] "+v"(stopr), [scur] "=&s"(scur), [st] "=&s"(st), [sn] "=&s"(sn), [rr] "=&v"(rr), [tm] "=&v"(tm), [u] "=&v"(u),
              [curn] "=&v"(curn)
            , [trips] "+v"(trips)
            : [ent2] "v"(ent2), [ntab] "v"(ntab), [base] "v"(base), [none] "v"(0xFFFFu), [se] "s"(se)
            : "vcc", "memory");
    } else {
        // the same chain through LDS instead of v_readlane: entry and zero table read with ds_bpermute (address in a VGPR,
        // no SGPR round trip)
        uint32_t a, b;
        asm volatile(
            "v_mov_b32_e32 %[stop], %[none]\n\t"
            "1:\n\t"
            "v_add_u32_e32 %[trips], 1, %[trips]\n\t"
            "v_lshlrev_b32_e32 %[a], 2, %[cur]\n\t"
            "ds_bpermute_b32 %[rr], %[a], %[ent2]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_bfe_u32 %[rr], %[rr], 6, 7\n\t"
            "v_add_u32_e32 %[t], %[zq], %[rr]\n\t"
            "v_min_u32_e32 %[tm], 63, %[t]\n\t"
            "v_lshlrev_b32_e32 %[b], 2, %[tm]\n\t"
            "ds_bpermute_b32 %[u], %[b], %[ntab]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add3_u32 %[curn], %[base], %[symbits], %[u]\n\t"
            "v_add_u32_e32 %[stopr], %[u], %[t]\n\t"
            "v_or3_b32 %[u], %[t], %[cur], %[u]\n\t"
            "v_cmp_gt_u32_e64 %[sok], 64, %[u]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[cur], %[cur], %[curn], %[sok]\n\t"
            "v_cndmask_b32_e64 %[zq], %[zq], %[t], %[sok]\n\t"
            "v_cndmask_b32_e64 %[stop], %[none], %[stopr], %[sok]\n\t"
            "v_cmp_gt_u32_e32 vcc, %[se], %[stop]\n\t"
            "s_cbranch_vccnz 1b\n\t"
            : [cur] "+v"(cur), [zq] "+v"(zq), [symbits] "+v"(symbits), [stop] "=&v"(stop), [sok] "+s"(sok), [t] "+v"(t), [stopr] "+v"(stopr),
              [rr] "=&v"(rr), [tm] "=&v"(tm), [u] "=&v"(u), [curn] "=&v"(curn), [a] "=&v"(a), [b] "=&v"(b)
            , [trips] "+v"(trips)
            : [ent2] "v"(ent2), [ntab] "v"(ntab), [base] "v"(base), [none] "v"(0xFFFFu), [se] "s"(se)
            : "vcc", "memory");
    }
    return cur + zq + symbits + bits + cv + stop + kprev;
}

// The chain on the SCALAR unit: the two v_readlane results are consumed by scalar instructions (a scalar instruction that
// reads an SGPR the vector unit has just written waits ~22 cycles), the windows behind the v_readlane are filled with the
// previous symbol's per-lane commits (vector) and with scalar work that does not need the result.  No wait states are owed
// (lane selects written by the scalar unit), nothing is decided by a branch but the loop.
__device__ __forceinline__ uint32_t run_scalar(uint32_t lane, uint32_t ent2, uint32_t ntab, uint32_t zq0, uint32_t se, uint32_t &trips) {
    auto shide = [](uint32_t v) {  // (see run(): no two operands may be known to hold the same value)
        uint32_t r;
        asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(v));
        return r;
    };
    auto vhide = [](uint32_t x) {
        asm volatile("" : "+v"(x));
        return x;
    };
    uint32_t cur = shide(0), zq = zq0, symbits = shide(0), kprev = shide(0), stop = shide(0xFFFFu), base = shide(0), ntrips = shide(0);
    uint32_t bits = vhide(0), cv = vhide(0), pl, sg = vhide(0), ze;
    uint32_t e, rr, t, tm, sn, adv, symn, stopr, curn, u, zf = shide(0);
    uint64_t spp, spb;
    asm volatile(
        "1:\n\t"
        "s_add_u32 %[ntrips], %[ntrips], 1\n\t"
        "v_readlane_b32 %[e], %[ent2], %[cur]\n\t"
        // previous symbol's per-lane commits (values in SGPRs written by the scalar unit: no hazard)
        "v_mov_b32_e32 %[pl], %[stop]\n\t"
        "v_or_b32_e32 %[pl], %[zf], %[pl]\n\t"
        "v_cmp_eq_u32_e64 %[spp], %[lane], %[pl]\n\t"
        "v_cmp_gt_u32_e64 %[spb], %[lane], %[kprev]\n\t"
        "v_mov_b32_e32 %[ze], %[symbits]\n\t"
        "s_bfe_u32 %[rr], %[e], 0x70006\n\t"
        "s_add_u32 %[t], %[zq], %[rr]\n\t"
        "s_min_u32 %[tm], %[t], 63\n\t"
        "v_readlane_b32 %[sn], %[ntab], %[tm]\n\t"
        "v_cndmask_b32_e64 %[cv], %[cv], %[sg], %[spp]\n\t"
        "v_cndmask_b32_e64 %[bits], %[bits], %[ze], %[spb]\n\t"
        "s_and_b32 %[adv], %[e], 63\n\t"
        "s_add_u32 %[symn], %[symbits], %[adv]\n\t"
        "s_and_b32 %[zf], %[e], 0x2000\n\t"
        "s_bfe_u32 %[u], %[e], 0x10000e\n\t"
        "v_mov_b32_e32 %[sg], %[u]\n\t"
        "s_or_b32 %[u], %[t], %[cur]\n\t"
        "s_add_u32 %[curn], %[symn], %[base]\n\t"
        "s_add_u32 %[stopr], %[sn], %[t]\n\t"
        "s_add_u32 %[curn], %[curn], %[sn]\n\t"
        "s_or_b32 %[u], %[u], %[sn]\n\t"
        "s_cmp_lt_u32 %[u], 64\n\t"
        "s_cselect_b32 %[cur], %[curn], %[cur]\n\t"
        "s_cselect_b32 %[zq], %[t], %[zq]\n\t"
        "s_cselect_b32 %[symbits], %[symn], %[symbits]\n\t"
        "s_cselect_b32 %[kprev], %[stopr], %[kprev]\n\t"
        "s_cselect_b32 %[stop], %[stopr], 0xffff\n\t"
        "s_cmp_lt_u32 %[stop], %[se]\n\t"
        "s_cbranch_scc1 1b\n\t"
        : [cur] "+s"(cur), [zq] "+s"(zq), [symbits] "+s"(symbits), [kprev] "+s"(kprev), [stop] "+s"(stop), [ntrips] "+s"(ntrips),
          [bits] "+v"(bits), [cv] "+v"(cv), [sg] "+v"(sg), [zf] "+s"(zf), [e] "=&s"(e), [rr] "=&s"(rr), [t] "=&s"(t), [tm] "=&s"(tm),
          [sn] "=&s"(sn), [adv] "=&s"(adv), [symn] "=&s"(symn), [stopr] "=&s"(stopr), [curn] "=&s"(curn), [u] "=&s"(u), [pl] "=&v"(pl),
          [ze] "=&v"(ze), [spp] "=&s"(spp), [spb] "=&s"(spb)
        : [ent2] "v"(ent2), [ntab] "v"(ntab), [lane] "v"(lane), [base] "s"(base), [se] "s"(se)
        : "scc", "memory");
    trips += ntrips;
    return cur + zq + symbits + bits + cv + stop + kprev;
}

// Round 6: the chain ENTIRELY on the scalar unit with its tables in SGPRs -- no v_readlane result is consumed on
// the chain.  The Huffman lookup is a static table in an SGPR block read with s_movrels_b32 (M0 = the next bits of the stream window,
// an s[2] pair shifted with s_lshl_b64); the block's history is an s[2] mask: the zero the symbol's run ends on by s_ff1_i32_b64 on
// the inverted mask behind the position, the correction bits owed for the non-zero coefficients passed by s_bfm_b64 + s_and_b64 +
// s_bcnt1_i32_b64; the new coefficient and its sign recorded with s_bitset1_b64 / s_lshl_b64 + s_or_b64 into a new-mask / sign-mask pair
// that the vector unit would apply once per block.  As in the other variants every entry is a plain symbol with run 0 (so the run
// loop -- one s_ff1 + s_bitset0 per skipped zero in a real stream -- makes no trip) and the window is never refilled: the FLOOR of
// such a chain.  The table here is 16 entries (one s[16] operand): a wave has 102 SGPRs in all, a 64-entry table would take 64 of them.
// (fixed registers: inline asm cannot name the halves of a 64-bit operand.  s[36:37] window, s[38:39] history, s[40:41] its
// inverse, s[42:43] new mask, s[44:45] sign mask, s[46:47] / s[48:49] / s[50:51] scratch pairs, s52.. scalars, s[64:79] the table)
__device__ __forceinline__ uint32_t run_sgpr_tables(uint32_t zq0, uint32_t se, uint32_t &trips, uint32_t seed) {
    uint32_t ntrips, res;
    asm volatile(
        "s_mov_b32 s36, 0x9abcdef0\n\ts_mov_b32 s37, 0x12345678\n\t"
        "s_mov_b32 s38, 0\n\ts_mov_b32 s39, 0x0f0f0000\n\t"
        "s_not_b64 s[40:41], s[38:39]\n\t"
        "s_mov_b64 s[42:43], 0\n\ts_mov_b64 s[44:45], 0\n\t"
        "s_add_u32 s52, %[zq0], 1\n\t"                    // k
        "s_mov_b32 s53, 0\n\t"                             // trips
        "s_mov_b32 s64, %[ent]\n\ts_mov_b32 s65, %[ent]\n\ts_mov_b32 s66, %[ent]\n\ts_mov_b32 s67, %[ent]\n\t"
        "s_mov_b32 s68, %[ent]\n\ts_mov_b32 s69, %[ent]\n\ts_mov_b32 s70, %[ent]\n\ts_mov_b32 s71, %[ent]\n\t"
        "s_mov_b32 s72, %[ent]\n\ts_mov_b32 s73, %[ent]\n\ts_mov_b32 s74, %[ent]\n\ts_mov_b32 s75, %[ent]\n\t"
        "s_mov_b32 s76, %[ent]\n\ts_mov_b32 s77, %[ent]\n\ts_mov_b32 s78, %[ent]\n\ts_mov_b32 s79, %[ent]\n\t"
        "1:\n\t"
        "s_add_u32 s53, s53, 1\n\t"
        "s_lshr_b32 s54, s37, 28\n\t"                      // the next 4 bits of the stream
        "s_mov_b32 m0, s54\n\t"
        "s_nop 0\n\t"
        "s_movrels_b32 s55, s64\n\t"                       // lut[m0]
        "s_and_b32 s56, s55, 31\n\t"                       // code length
        "s_bfe_u32 s57, s55, 0x40008\n\t"                  // run
        "s_lshl_b64 s[36:37], s[36:37], s56\n\t"           // the code is consumed
        "s_lshr_b64 s[46:47], s[40:41], s52\n\t"           // zeros at and behind the position
        "s_cmp_eq_u32 s57, 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "2:\n\t"                                           // skip r zeros (no trip here: run 0)
        "s_ff1_i32_b64 s58, s[46:47]\n\t"
        "s_bitset0_b64 s[46:47], s58\n\t"
        "s_sub_u32 s57, s57, 1\n\t"
        "s_cmp_lg_u32 s57, 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "3:\n\t"
        "s_ff1_i32_b64 s58, s[46:47]\n\t"                  // the zero the run ends on, relative to k
        "s_add_u32 s59, s52, s58\n\t"                      // pos
        "s_bfm_b64 s[48:49], s58, s52\n\t"                 // coefficients k .. pos - 1
        "s_and_b64 s[48:49], s[48:49], s[38:39]\n\t"
        "s_bcnt1_i32_b64 s60, s[48:49]\n\t"                // correction bits owed
        "s_lshl_b64 s[36:37], s[36:37], s60\n\t"
        "s_lshr_b64 s[50:51], s[36:37], 63\n\t"            // the new coefficient's sign
        "s_lshl_b64 s[36:37], s[36:37], 1\n\t"
        "s_bitset1_b64 s[42:43], s59\n\t"
        "s_lshl_b64 s[50:51], s[50:51], s59\n\t"
        "s_or_b64 s[44:45], s[44:45], s[50:51]\n\t"
        "s_add_u32 s52, s59, 1\n\t"
        "s_cmp_lt_u32 s52, 63\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b32 %[ntrips], s53\n\t"
        "s_add_u32 %[res], s42, s45\n\t"
        : [ntrips] "=s"(ntrips), [res] "=s"(res)
        : [zq0] "s"(zq0), [ent] "s"(1u | (seed & 0u))
        : "scc", "m0", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52",
          "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76",
          "s77", "s78", "s79");
    trips += ntrips;
    return res;
}

template <int kVariant>
__global__ void k(uint64_t *out, uint32_t seed) {
    const uint32_t lane = threadIdx.x;
    // every window entry: 0 code bits (so the window offset stays put), run 0, not ZRL; the zero table: rank r -> 0 (offset stays 0)
    const uint32_t ent2 = (1u << 6) | (seed & 0u), ntab = 0;
    uint32_t sink = 0;
    uint64_t dt[2];
    uint32_t trips[2] = {0, 0};
    for (int warm = 0; warm < 2; warm++)
        for (int which = 0; which < 2; which++) {
            const uint32_t zq0 = which == 0 ? 31u : 0xFFFFFFFFu;  // 32 trips or 64
            const uint64_t t0 = tick();
            for (int rep = 0; rep < 64; rep++)
                sink += kVariant == 4   ? run_sgpr_tables(__builtin_amdgcn_readfirstlane(zq0), 200u, trips[which], seed)
                        : kVariant == 3 ? run_scalar(lane, ent2, ntab, __builtin_amdgcn_readfirstlane(zq0), 200u, trips[which])
                                        : run<kVariant>(lane, ent2, ntab, zq0 + (sink & 0u), 200u, trips[which]);
            dt[which] = tick() - t0;
        }
    if (lane == 0) {
        out[0] = dt[0];
        out[1] = dt[1];
        out[2] = sink;
        out[3] = trips[0];
        out[4] = trips[1];
    }
}

int main() {
    uint64_t *d, h[5];
    hipMalloc(&d, sizeof h);
    const char *names[5] = {"full loop (as shipped)", "chain only, gaps as s_nop", "chain through ds_bpermute", "chain on the scalar unit",
                            "scalar chain, tables in SGPRs"};
    for (int v = 0; v < 5; v++) {
        if (v == 0) k<0><<<1, 64>>>(d, 5);
        if (v == 1) k<1><<<1, 64>>>(d, 5);
        if (v == 2) k<2><<<1, 64>>>(d, 5);
        if (v == 3) k<3><<<1, 64>>>(d, 5);
        if (v == 4) k<4><<<1, 64>>>(d, 5);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        // 64 entries of (fixed cost + n trips); the trip counts differ by 32 (+ the one that fails: the same in both)
        // two timed passes of 64 loop entries each (the counters run over both)
        printf("%-32s %7.2f ticks per trip (short entries: %llu ticks, %llu trips; long entries: %llu ticks, %llu trips)\n", names[v],
               (double)(h[1] - h[0]) / ((double)(h[4] - h[3]) / 2.0), (unsigned long long)h[0], (unsigned long long)h[3] / 2,
               (unsigned long long)h[1], (unsigned long long)h[4] / 2);
    }
    return 0;
}
