#!/usr/bin/env python3
"""Generates lzma_rs_amd/csrc/fast_loop_asm.inc: the LZMA symbol loop of the fast kernel as ONE gfx950
inline-asm statement per variant (text + operand lists), so that nothing in the hot loop is left to hipcc's
register allocator / block placement (rocprof on a C++ loop: 45 % of issued instructions were phi copies and
other compiler glue).  Variants, all with the same register numbering (one kernel, 128 VGPRs, four waves per SIMD): LP0 (lp == 0,
pb <= 2: the headline), GEN (any lp, pb <= 2), PB4 (pb 3 / 4) -- these three for lc + lp <= 3 -- and HBM (lc + lp >= 4, any pb: the
literal rows in a slab in memory behind eight cached rows; round 4 -- it replaced the LC4 variant and its 152-VGPR kernel, which ran
lc + lp = 4 at three waves per SIMD: 11.9 GB/s against 15.1 now, profiles/r04_lclp_classes.txt) -- and, round 6, HB0: the same slab and
row caches for lc >= 4 with lp == 0 and pb <= 2 (lc4/lp0/pb2, the one such class liblzma's presets reach) without the GEN / PB4
variants' dearer bookkeeping: 56.5 instead of 58.2 instructions per byte on lc4 text (the LP0 loop on lc3: 55.2).

What the loop does is DecoderState::process_mode(Finish) (src/decode/lzma.rs:435-524) with
decode_literal (526-561), decode_distance (563-592), LenDecoder::decode (rangecoder.rs:256-269),
RangeDecoder::decode_bit / get_bit / normalize (rangecoder.rs:59-120) and the short-match part of
LzCircularBuffer::append_lz (lzbuffer.rs:255-281).  Everything rare leaves the loop with an exit code
and is finished by the C++ around it (decode_fast_asm.hip.h): errors, long or clipped matches, the
end-of-stream marker, the output limit.

Every change here is checked on the CPU first: tools/emu executes the generated text instruction by
instruction (tests/test_asm_emulator.py) and counts what runs (tools/emu/profile.py).

Conventions inside the loop
  * range and code are the aligned SGPR pair s[66:67]; every other piece of LZMA state is an SGPR too; the
    probability model is lane-resident (one probability per lane of a VGPR, see decode_fast_asm.hip.h).
  * A decision, form A (single decisions: is_match, is_rep ..., literal levels 6-7, matched literals): every
    lane computes the bound of its own probability, (range >> 11) * p (v_lshrrev, v_mul_u32_u24), v_readlane
    picks the node's; s_sub of code - bound sets SCC = (code < bound) = "bit is 0"; two s_cselect pick
    range / code.  Form B (walks of a tree whose update is deferred): the vector ALU also delivers
    range - bound, two v_readlane fill scalar pairs and ONE s_cselect_b64 picks (bound, code) or
    (range - bound, code - bound): 3 scalar + 5 vector instructions, no wait state.  No EXEC manipulation anywhere.
  * What the hardware charges (measured at full occupancy, profiles/r02_*): a scalar instruction costs 2.5x a
    vector one; a DEPENDENT vector instruction on the decision chain about as much as two scalar ones; a taken
    branch next to nothing.  Hence:
      - the probabilities of a walked tree are updated once per walk (tree_update): the final symbol names the
        visited node of every level and the bit decided there, so every lane can tell from its own index
        whether it was visited; single decisions update their lane under a v_cmp_eq(lane) / v_cndmask mask,
        on the side of the branch that knows the bit where there is one;
      - rare conditions share a guard: one compare at the top of a symbol (gtop: size reached / reader may be at
        EOF / within 273 bytes of the output limit), one per match (gdist: distance beyond min(len, dict_size), end
        marker, length >= 64, near the output limit);
      - the pos_slot walk's final symbol indexes two per-lane tables: the distance base and the ADDRESS of the code
        that continues this slot (tables_prologue); one s_setpc_b64 enters the chain of 26 direct-bit blocks so that
        exactly the needed number run.
  * The SIMD's arbiter serves its waves oldest first, which starves the youngest of four equally busy waves (and the
    kernel ends with the slowest): every window refill sets s_setprio (((s_memtime >> 21) + wave slot) & 3), so the
    four waves of a SIMD take turns at every priority and finish within 2 % of each other.
  * Symbols are accumulated with s_addc, i.e. with INVERTED bits (SCC = bit is 0).  Tree nodes are
    therefore stored at the lane of the inverted path, which is a permutation inside each tree level
    and costs nothing (all probabilities start equal); decoded values are un-inverted once per symbol.
  * Trees whose first node is not lane 1 use heap numbering from another root (root r: children 2r,
    2r+1), so that the running symbol IS the lane and no per-bit address add is needed.
  * Normalisation (one per input byte) is an out-of-line stub per site behind s_cmp_lt_u32 + s_cbranch_scc1.
  * Input: a 64-byte window, one byte per lane (winb), the next one prefetched (winb_next); `off` is
    the lane of the next byte, `lim` the value of `off` at which the decoder's reader is at EOF.
  * Emission lists: main (hot fall-through path), cold (out of line), cold2 (out-of-line code of code that is itself
    in cold), stubs (normalisation / mismatch stubs); concatenated in that order.
  * gfx940-family wait states hipcc would insert but inline asm must provide itself: one between a VALU write
    of a VGPR and a v_readlane of it (measured: without it every stream decodes wrongly), two between a VALU
    write of VCC and a VALU read of it (tests/test_host_abi.py lints the generated text for both).
"""
import os
import re

PEND_UNKNOWN = 0x100  # pend_n value meaning "no pending match, but prev / mb are not at hand"
LOAD_MOD = os.environ.get("MILZMA_GEN_LOAD_MOD", "")    # cache-policy bits of the match-source load (experiments)
STORE_MOD = os.environ.get("MILZMA_GEN_STORE_MOD", "")  # ... of the literal / match stores
WAITPROF = os.environ.get("MILZMA_GEN_WAITPROF", "0") == "1"   # tuning: s_memtime around the two s_waitcnt vmcnt(0) sites
WAITPROF2 = os.environ.get("MILZMA_GEN_WAITPROF", "0") == "2"  # tuning: isolated round trips of a literal's byte store / a match's load
# sensitivity probes (tuning only): k dead scalar / vector / never-taken-branch instructions per adaptive decision
PAD_S = int(os.environ.get("MILZMA_GEN_PAD_S", "0"))   # (writes s96: free unless ALIGNLAZY / WAITPROF)
PAD_V = int(os.environ.get("MILZMA_GEN_PAD_V", "0"))
PAD_B = int(os.environ.get("MILZMA_GEN_PAD_B", "0"))
# wave priority rotation: the SIMD's issue arbiter serves the oldest wave first, which (measured, per-wave clocks) lets the
# first-dispatched wave of a SIMD run at lone-wave speed (222 ms) while the youngest needs 324 ms -- and the kernel ends
# with the youngest.  PRIO = k > 0: every window refill sets s_setprio ((len >> k) + wave slot) & 3, so that the four waves
# of a SIMD take turns at every priority and finish together.  PRIO = -1: static priority = wave slot (diagnostic).
PRIO = int(os.environ.get("MILZMA_GEN_PRIO", "12"))
# PRIO_LAST: rotate only once the launch's last block has started (a flag it sets, polled at every window refill): while
# blocks are still waiting for a slot, waves that finish staggered (oldest first) free slots early and the CU stays full
# -- 6144 streams (1.5 rounds): 14.2 GB/s without rotation, 12.9 with; for one round (<= 4096 streams) the flag is up at once.
PRIO_LAST = os.environ.get("MILZMA_GEN_PRIO_LAST", "1") == "1"
PRIO_TIME = int(os.environ.get("MILZMA_GEN_PRIO_TIME", "21"))  # k > 0: rotate on the shader clock ((s_memtime >> k) + slot) instead of on len
# Decision "form B": range and code live in the adjacent pair s[66:67]; the vector ALU also delivers range - bound, both are
# read into scalar pairs and ONE s_cselect_b64 picks (bound, code) or (range - bound, code - bound): 3 scalar + 5 vector
# instructions and no wait state, against 5 + 3 + s_nop for form A.  Measured marginal cost (dead-instruction probes, r02):
# a scalar instruction per byte costs 3x what a vector instruction does.  FORMB: comma list of where it is used
# (tree = deferred-update tree walks, single = is_match / is_rep / choice ..., lit = literal levels 6-7 and matched literals).
FORMB = set(filter(None, re.split("[,+]", os.environ.get("MILZMA_GEN_FORMB", "tree"))))
# range >> 11 on the scalar ALU (s_lshr_b32, then v_mul with a scalar operand) instead of as a dependent vector instruction:
# list of site kinds (tree, single, lit)
R11S = set(filter(None, re.split("[,+]", os.environ.get("MILZMA_GEN_R11S", ""))))
# the "range < 2^24" test on the scalar ALU (s_cmp_lt_u32 + s_cbranch_scc1) instead of the vector ALU (v_cmp + s_cbranch_vccnz):
# list of site kinds (tree, single, lit, direct), or 1 = everywhere
NORM_S = set(filter(None, re.split("[,+]", os.environ.get("MILZMA_GEN_NORM_S", "1"))))
# normalisation stub: one s_lshl_b64 of the (range, code) pair instead of two 32-bit shifts (range's top byte is zero there)
NORM64 = os.environ.get("MILZMA_GEN_NORM64", "1") == "1"
# Shadow scheduling (round 3, experiments/microbench/shadow_slots.hip): the first scalar instruction after a v_readlane issues
# ~19 cycles after it; up to four independent VECTOR instructions placed between the v_readlane(s) of a decision and its s_sub are
# free for a lone wave and cost 5.4 cycles instead of 10 at 4 waves per SIMD (scalar ones gain nothing there).  DEFER: which
# probability updates are queued and emitted into the NEXT decisions' shadows instead of after their own decision:
#   single = the update of is_match / is_rep / choice ... decisions and of immediate-update tree levels (3 instructions),
#   tree   = the once-per-walk update of the literal / pos_slot trees (7 instructions).  SHADOW: instructions per shadow.
# Direct bits without normalisation tests: where the normalisations of a run of n direct bits fall follows from the leading zeros
# of `range` at its start (the first after 8 - clz halvings, then every 8), so eight pre-laid chains (one per (n + clz) mod 8) carry
# unconditional inline normalisations at fixed places and 4-instruction bit blocks; the entry address is computed per lane on the
# vector ALU from tbl_b and clz and fetched with the one v_readlane that fetched the table entry before.
DIRECT8 = os.environ.get("MILZMA_GEN_DIRECT8", "1") == "1"
# The state after a literal (lzma.rs:472-478) from a per-lane table with one v_readlane instead of two / three scalar instructions
# (verdict item 1a: "state transition by table"); the v_readlane sits in front of the literal walk's vector instructions, so the
# vector -> scalar hop it starts is over before the next scalar instruction wants to issue.
# `off` (lane of the next input byte) and `lim` are kept biased by -64 inside the loop: the increment after a byte then carries out exactly
# when the window is used up, and the carry is the refill test (no s_bitcmp); v_readlane takes the lane from the low 6 bits either way.
OFFBIAS = os.environ.get("MILZMA_GEN_OFFBIAS", "1") == "1"
# EOFWRAP (needs OFFBIAS): the last window of the input is loaded END-aligned (its lane 63 is the last byte), so that "window used up"
# and "reader at EOF" are the same carry: the normalisation stubs lose their `off == lim` test.  lim (biased) = bytes beyond the
# current window; -1 = nothing left at all (then off = -1: the next byte asked for carries at once and is the EOF error).
# Invariant everywhere: reader position = wbase + 64 + off (biased).
EOFWRAP = OFFBIAS and os.environ.get("MILZMA_GEN_EOFWRAP", "1") == "1"
# PINV = first VGPR of a fixed block for the model registers the compiler otherwise places (OPS_INOUT_V: 35 registers); 0 = its choice.
# (The same loop ran 33 % slower inside the time-sliced kernel than inside the ordinary one at 4 waves per SIMD, identical text,
#  different register numbers: with the block pinned both kernels run the loop on the same registers.)
# Round 4, alternating A/B on the bench batch (profiles/r04_kernel_ab.txt): PINV=20 -0.4 %, with LITSPLIT -0.7 %: both are the default now.
PINV = int(os.environ.get("MILZMA_GEN_PINV", "20"))
# LITSPLIT: the two dwords of a plain literal row r at v[64 + r] and v[64 + rows + r] instead of v[64 + 2r], v[65 + 2r]: the gpr index
# of a row is the row itself, not twice it -- two scalar shifts less per row swap (0.14 per byte on text, 1.75 on random data).
# Round 4: timed (see PINV above), default on.
LITSPLIT = os.environ.get("MILZMA_GEN_LITSPLIT", "1") == "1"
# VBASE: first register of the loop's fixed VGPR block (literal table, pos_slot trees, temporaries and per-lane constants: 56 registers,
# 72 for lc + lp = 4).  64 (the default) puts the block's top at v119.  (Round 3 also had VROW8 / NOPB4 for a five-waves-per-SIMD build of
# the 8-row variants: +2.9 % for >= 5120 streams, nothing for 4096 -- removed in round 4, profiles/r03_kernel_ab.txt.)
# SWAP2 (round 4): a row swap packs the walked row straight into its indexed registers and unpacks the new one straight out of them (11
# instructions instead of 16: no staging moves).  DISPMAD: the entry into the direct-bit chains by two v_mad instead of two v_mul + add + sub
# (the chain stride and the normalisation block's size as per-lane constants).
SWAP2 = os.environ.get("MILZMA_GEN_SWAP2", "1") == "1"
DISPMAD = os.environ.get("MILZMA_GEN_DISPMAD", "1") == "1"
# DISP2 (needs DISPMAD): entry = B2 + clz * stride - ((n + clz) >> 3) * (8 * stride + normalisation size) with B2 = offset + n * stride
# in tbl_b's upper 24 bits: three dependent vector instructions in front of the v_readlane instead of six, and tbl_a's v_readlane next to
# the entry's (one vector -> scalar hop for both).
DISP2 = os.environ.get("MILZMA_GEN_DISP2", "1") == "1"
# SSHADOW: scalar bookkeeping that nothing waits for (state after a literal / a match, the rep rotation, the pos_slot tree's way back to
# its register) is emitted in the NEXT decision's v_readlane shadow, where the wave would otherwise wait out the vector -> scalar hop
# (a scalar instruction costs ~3 cycles there instead of ~6, experiments/microbench/shadow_slots.hip).  EARLYLDS: a matched literal's
# row is requested from LDS as soon as the row is known, waited for where it is first used.
SSHADOW = set(filter(None, re.split("[,+]", os.environ.get("MILZMA_GEN_SSHADOW", ""))))   # of: state, rep, wb
# NBPRE (needs EOFWRAP): the next input byte is read ahead into an SGPR (nb).  A normalisation shifts it in at once and reads the one
# after it as its LAST instruction, so that v_readlane's vector -> scalar hop runs under the next decision's vector head instead of in
# front of the stub's own scalar instructions (micro-benchmark: 80 -> 69 cycles per normalisation at four waves per SIMD).
NBPRE = os.environ.get("MILZMA_GEN_NBPRE", "1") == "1"
# LENDEFER (not for 16 position states): the update of a new match's low length tree (the common lengths 2..9) rides in the shadows of the
# pos_slot walk's first two decisions instead of standing between the two walks (seven vector instructions, five of them dependent).  The
# rare lengths (mid / high tree) used to join in front of the pos_slot walk; they get their own copy of its first two levels and join behind.
LENDEFER = os.environ.get("MILZMA_GEN_LENDEFER", "0") == "1"   # (measured, profiles/r04_kernel_ab.txt section 4: nothing -- off)
# ALIGNLAZY (with LENDEFER's split): nothing follows the align walk that could carry its tree's update in a shadow, so the walk only
# remembers its final symbol (asym); the update is made in the shadows of the NEXT new match's pos_slot walk (levels 2 and 3) -- the align
# tree is read nowhere else -- or when the loop is left.  asym = 0: nothing pending (lane 0 of m_align is no node).
ALIGNLAZY = os.environ.get("MILZMA_GEN_ALIGNLAZY", "0") == "1" and not (WAITPROF or WAITPROF2)   # (the wait profiles use asym's register)
EARLYLDS = os.environ.get("MILZMA_GEN_EARLYLDS", "1") == "1"
# MLGUARD: a matched literal's two distance checks (lzma.rs:541-546) behind the match guard gdist (<= min(len, dict_size)): one compare
# instead of three; the exact checks out of line.
MLGUARD = os.environ.get("MILZMA_GEN_MLGUARD", "1") == "1"
# Round 5 (profiles/r05_kernel_ab.txt; the pipes' prices: profiles/r05_pipe_peaks.txt -- a v_readlane occupies the vector pipe for 8 cycles, a
# multiply / VOP3 / scalar-operand instruction for 4, a scalar instruction the CU's scalar pipe for 4 of the SIMD's cycles):
#   S1: list of site kinds (single) whose decision is all scalar -- the node's probability by v_readlane, s_lshr + s_mul_i32 for the bound:
#       1 vector + 6 scalar instructions instead of 3 + 4 ("mixed forms that balance the pipes")
#   FORMA2: the deferred-update tree walks in form A (one v_readlane, range - bound and both selects on the scalar ALU: 3 vector + 6 scalar
#       instead of 5 + 4) WITH the shadows form B has (the old form A path flushed the queue and carried an s_nop)
#   K24S: the constant 2^24 of the normalisation tests in an SGPR (4-byte s_cmp instead of 8-byte: 3.2 tests per byte)
#   SYM_M0: the running symbol of a tree walk lives in m0 -- it is the lane select of the walk's v_readlane, and a v_readlane whose lane comes from
#       an SGPR occupies the vector pipe for 8 cycles where one with a constant lane takes 4 (profiles/r05_pipe_peaks.txt); m0 is not read
#       through the SGPR file.  (s_set_gpr_idx_on writes m0: the two places where the symbol was still live across one are re-ordered.)
SYM_M0 = os.environ.get("MILZMA_GEN_SYM_M0", "1") == "1"   # (round 5: 238.3 -> 222.4 ms, profiles/r05_kernel_ab.txt)
SYM_M0 = SYM_M0 and "wb" not in SSHADOW and not ALIGNLAZY and not LENDEFER   # (round 4's rejected knobs keep the symbol, or a queued
#                                                                              update that reads it, live across an s_set_gpr_idx_on)
S1 = set(filter(None, re.split("[,+]", os.environ.get("MILZMA_GEN_S1", ""))))
FORMA2 = os.environ.get("MILZMA_GEN_FORMA2", "0") == "1"
#   PRESYMV: the update constant of an immediately updated tree level (K = 2048 for a 0 bit, 31 for a 1 bit) from the symbol's new low bit on the
#       vector ALU (v_and + v_mad, queued with the update) instead of s_cselect_b32 behind the decision: one scalar instruction less per such level
PRESYMV = os.environ.get("MILZMA_GEN_PRESYMV", "0") == "1"
#   ALIGN_STUBS: log2 of the alignment of out-of-line branch targets that are only ever reached by a taken branch (the normalisation stubs: 0.74 taken
#       branches per byte go in and out of them) -- the padding in front of such a label is never executed
ALIGN_STUBS = int(os.environ.get("MILZMA_GEN_ALIGN_STUBS", "0"))
#   ALIGN_FREE (round 6): log2 of the alignment of EVERY label that cannot be fallen into (the instruction in front of it is an unconditional branch /
#       s_setpc): the padding is never executed, so taken-branch targets -- the loop top after a literal, `match`, the rep branches, the stubs, the
#       out-of-line paths -- start a fetch line for free.  (Not inside the direct-bit chains: their entries are computed from equal strides.)
ALIGN_FREE = int(os.environ.get("MILZMA_GEN_ALIGN_FREE", "0"))
K24S = os.environ.get("MILZMA_GEN_K24S", "0") == "1"
#   VDIRECT (round 6): the direct-bit chains on the VECTOR ALU, on (range, code, accumulator) in VGPRs that every lane holds alike: per bit
#       v_lshrrev range / v_addc acc (the previous bit's borrow) / v_sub_co / v_min -- four VOP2 instructions on vector registers only where
#       the scalar form takes four scalar instructions.  min(code, code - range) IS the new code (the difference wraps above code exactly
#       when code < range), so only the accumulator reads vcc -- two instructions behind the v_sub_co that wrote it (gfx940: a VALU read of an
#       SGPR a VALU wrote needs two wait states).  Into the chain: three moves; out of it: the last borrow and three v_readfirstlane.
#       MEASURED (profiles/r06_kernel_ab.txt section 3): nothing -- v_sub_co, v_min and v_addc are 4.3-cycle forms like the scalar pipe's 4.1
#       (experiments/microbench/pipe_peaks.hip: the vector block 16.7 cycles per bit and SIMD at four waves, the scalar one 16.5), and moving
#       a sixth of the scalar instructions to the other pipe, all of them or any share, leaves the kernel where it was.  Off.
#       VDIRECT_S: the LAST so many bits of every chain stay on the scalar ALU (the pipes' loads meet somewhere in between): the chain then crosses
#       from the vector registers to (range, code, t4) in front of them, else at its end.
VDIRECT = os.environ.get("MILZMA_GEN_VDIRECT", "0") == "1"
#   QDIRECT = j0 (round 6; 0 = off): behind a normalisation inside a direct-bit chain the range is r << 8 and the next bits are floor(code / (range >> j))
#       EXACTLY while code < range (get_bit's halvings shift out zeros only) -- j = min(bits up to the next normalisation or the chain's end, 6) of them at once
#       when j >= j0: lane L tests the candidate c = (2^j - 1 - L) & (2^j - 1), code - c * (range >> j) < (range >> j) (no product wraps: c < 2^j,
#       range >> j < 2^(32 - j)); the first lane that answers IS the inverted quotient the accumulator wants; no lane answers when code >= range (a damaged
#       stream): then the serial bits run.  7 scalar + 4 .. 5 vector instructions and 2 + 2 branches against 4 j scalar ones.  The block lives out of line; every normalisation block of the
#       chains carries one more instruction (the branch to it, or s_nop): the entry arithmetic stays as it is.
#       Measured (profiles/r06_kernel_ab.txt section 6): from 5 / 4 / 3 / 2 bits on -1.2 / -1.5 / -1.75 / -1.7 % at dict 64 KiB, -2.6 % at 8 MiB.  3 ships.
QDIRECT = int(os.environ.get("MILZMA_GEN_QDIRECT", "3"))
VDIRECT_S = int(os.environ.get("MILZMA_GEN_VDIRECT_S", "0"))
VBASE = int(os.environ.get("MILZMA_GEN_VBASE", "64"))
# Where the loop's code falls in the 32-byte instruction fetch lines is worth +-1.2 % (round 5, profiles/r05_kernel_ab.txt section 4: the loop 64-byte
# aligned + 0 / 2 dwords 224.5 ms, + 4 / 6 dwords 222.4, + 8 / 10 225.0, + 12 / 14 223.0: period 32 bytes).  Left to wherever the compiler's code in front
# of the asm statement ends, the phase changed with every edit of the C++ around the loop; pinned: 32-byte aligned + 6 dwords (221.8 ms; + 4: 222.0, + 5: 222.5, + 7: 223.4; unpinned as it happened to fall: 221.8).
ALIGN = int(os.environ.get("MILZMA_GEN_ALIGN", "5"))   # log2 of the alignment of the loop's first instruction (0: wherever the compiler's code ends)
ALIGN_PAD = int(os.environ.get("MILZMA_GEN_ALIGN_PAD", "3"))   # ... + this many 4-byte s_nop behind the alignment: the phase of the loop's code in its fetch lines
STATE_TBL = os.environ.get("MILZMA_GEN_STATE_TBL", "0") == "1"   # (measured: 0.8 % slower on text, 3 % on random data -- off)
DEFER = set(filter(None, re.split("[,+]", os.environ.get("MILZMA_GEN_DEFER", "single,tree"))))
SHADOW = int(os.environ.get("MILZMA_GEN_SHADOW", "6"))   # (round 5: 6 instead of 4: -0.4 %, with and without SYM_M0)
if "1" in NORM_S:
    NORM_S = {"tree", "single", "lit", "direct"}


# ---- physical temporaries (listed as clobbers; never live across the asm statement) ----------------
S = dict(sp="s72", sb="s73", sr1="s74", sc1="s75", sk="s76", ln="s77", sym="s78", n0="s79", n1="s80",
         ps="s81", row="s82", t0="s83", t1="s84", t2="s85", t3="s86", t4="s87", t5="s88", t6="s89",
         pad="s96", nb="s69", asym="s96", st="s72", prioph="s68", jb_lo="s64", jb_hi="s65", pl0="s97", gtop="s70", gdist="s71",
         clk_lo="s94", clk_hi="s95", clk_t="s96")  # s[94:95] / s96: s_memtime of the priority rotation and of the wait profiles
S["lnm"] = S["ln"]   # the lane of the is_match decision: no tree walk is in progress at a symbol's top, so m0 is free for it too
if SYM_M0:
    S["sym"] = S["lnm"] = "m0"
MPAIR = "s[98:99]"  # a second lane mask
DM = "s[90:91]"     # lane mask of a deferred tree update (the constants 2017 / 2048 that used to live there are VGPRs now)
JPAIR, JPAIR_LO, JPAIR_HI = "s[98:99]", "s98", "s99"  # target of the computed jump into the direct-bit chain
JBASE = "s[64:65]"  # address of Lbase (set once per entry): jump targets are table offsets from it
RET = "s[92:93]"  # return address of the window refill subroutine
_V0 = dict(M0=84, M1=85, M2=86, M3=87, VT0=88, VT1=89, VT2=90, VA=91, VPS=92, vt=93, VR=94, VL16=95, VOOB=96, VKTOP=97, vx=98,
           VLANE64=99, VLANE128=100, VLANE192=101, vb=102, VSH6=103, VSH6M1=104, VSH5=105, VSH5M1=106, VSH4=107, VSH4M1=108,
           VLEVEL=109, va=110, vr=112, VLANEM1=113,
           DVT=114, DVA=115, DVX=116, c2017=117, c2048=118, VSTT=119, VCH=111, VNDN=119, VB2=97)   # (VSTT only with STATE_TBL, vpad only with PAD_V)  # temporaries of deferred updates; the constants 2017 and 2048
NBPRE = NBPRE and EOFWRAP
if PAD_V:
    _V0["vpad"] = _V0["VT2"]        # (never live across a decision: the dead-instruction probes leave the loop itself as it ships)
if STATE_TBL:
    DISPMAD = False
# (VB2 = tbl_b >> 8 lives in VKTOP's register, which is free when every range test is scalar; with a vector-side test it is recomputed where
#  it is used: one vector instruction per match)
DISP2 = DISP2 and DISPMAD and DIRECT8
VDIRECT = VDIRECT and DISP2 and NBPRE
QDIRECT = QDIRECT if (DISP2 and NBPRE and not VDIRECT) else 0
VB2_INLINE = DISP2 and not NORM_S >= {"tree", "single", "lit", "direct"}
V, MROW, PS0, PS0M2, CLOBBER_V, LIT_REGS = {}, "", "", "", [], 16
LIT0, LIT1 = "v%d" % VBASE, "v%d" % (VBASE + 1)   # literal plain table: 2 dwords per row from v64 (fixed, indexed with s_set_gpr_idx)


def set_layout():
    """Fixed VGPR numbering: the plain literal table (eight rows, 16 dwords) at v64.., then the four pos_slot trees (PS0..), then the
    temporaries and per-lane constants."""
    global V, MROW, PS0, PS0M2, CLOBBER_V, LIT1
    LIT1 = "v%d" % (VBASE + LIT_REGS // 2) if LITSPLIT else "v%d" % (VBASE + 1)
    V.clear()
    off = VBASE - 64
    V.update({k: "v%d" % (n + off) for k, n in _V0.items()})
    MROW = "v[%d:%d]" % (84 + off, 87 + off)
    PS0 = "v%d" % (80 + off)        # pos_slot trees for len_state 0..3
    PS0M2 = "v%d" % (78 + off)      # PS0 - 2: indexed with len_state + 2
    CLOBBER_V = sorted(set(V.values()), key=lambda r: int(r[1:]))


set_layout()
CLOBBER_S = sorted((set(S.values()) | {"s90", "s91", "s92", "s93", "s98", "s99"}) - {"m0"}, key=lambda r: int(r[1:]))   # (m0 cannot be listed: the compiler sets it wherever it needs it)  # (+ DM, the refill return address)

EXIT = dict(DONE_SIZE=0, DONE_FIN=1, INPUT_EOF=2, MARKER=3, LIMIT=4, LZ_SLOW=5, MATCH_DIST_DICT=6,
            MATCH_DIST_OUT=7, LZ_DIST_DICT=8, LZ_DIST_OUT=9, QUANTUM=10, NEED_INPUT=11)
# FEED (round 5, MILZMA_DECODE_FEED): bit 5 of the lc8 operand (the shift that uses it takes the low five bits) says that the input the reader
# sees is a VIEW that will be continued: the loop then leaves at the first symbol top with fewer than FEED_MARGIN bytes of the view left
# (exit NEED_INPUT: nothing of the next symbol looked at; the reference's Stream keeps 20 bytes back for the same reason, stream.rs) instead of
# running a symbol into the end of the view.  Costs the ordinary loop two scalar instructions per window refill (set_guards).
FEED = os.environ.get("MILZMA_GEN_FEED", "1") == "1"   # (0: the loop without it -- A/B of what its few scalar instructions and the shift of the code behind them cost)
FEED_BIT = 5
FEED_GUARD = 1 << FEED_BIT   # symbol tops look closer (the slow way round) once fewer than this many bytes lie beyond the reader's window
FEED_MARGIN = 20             # ... and leave when fewer than this many bytes of the view are left: the most one symbol can read (the reference's
                             # MAX_REQUIRED_INPUT, lzma.rs:13 -- its streaming mode decodes a symbol without a trial run on exactly this condition)
# ---- operands -------------------------------------------------------------------------------------------
OPS_INOUT_S = ["range", "code", "off", "lim", "wbase", "len", "state", "rep0", "rep1", "rep2", "rep3", "prev",
               "mb", "pend_n", "pend_pos", "cur_row", "mlen", "exitcode", "prof_wm", "prof_nm", "prof_wc", "prof_nc", "tbl_ready"]
OPS_INOUT_V = ["m_ismatch", "m_rep", "m_rep0long", "m_align", "m_posdec_a", "m_posdec_b", "m_len_low", "m_len_mid",
               "m_len_h0", "m_len_h1", "m_len_h2", "m_len_h3", "m_rlen_low", "m_rlen_mid", "m_rlen_h0", "m_rlen_h1",
               "m_rlen_h2", "m_rlen_h3", "u0", "u1", "u2", "u3", "winb", "winb_next", "pend_val", "tbl_a", "tbl_b",
               "vtag", "vtagm",   # hbm variant: whose row each of the eight register / LDS row slots holds (lanes 0..7; -1: none)
               # pb 3 / 4 (PB4 variant): position states 4..15
               "m_ismatch_b", "m_ismatch_c", "m_rep0long_b", "m_rep0long_c", "m_len_low_b", "m_len_mid_b", "m_rlen_low_b",
               "m_rlen_mid_b"]
PB4_ONLY_V = ["m_ismatch_b", "m_ismatch_c", "m_rep0long_b", "m_rep0long_c", "m_len_low_b", "m_len_mid_b", "m_rlen_low_b", "m_rlen_mid_b"]
OPS_IN_S = ["out_lim", "safe_len", "target", "qtop", "known", "dict_size", "lc", "lc8", "lpmask", "pbmask", "in_rsrc",
            "out_rsrc", "ldsbase", "flagptr", "lit_rsrc"]   # lit_rsrc: the literal rows' slab in HBM (hbm variant: lc + lp > 4)
OPS_IN_V = ["v_lane"]
FIXED_OPERANDS = {"range": "s66", "code": "s67"}   # an aligned pair, for s_cselect_b64
RC = "s[66:67]"
RC1 = "s[74:75]"                                   # (sr1, sc1)


def R(name):
    if name in S:
        return S[name]
    if name in V:
        return V[name]
    if name in FIXED_OPERANDS:
        return FIXED_OPERANDS[name]
    return "%%[%s]" % name


def role(name):
    """decorator: the instructions a Gen method emits play this role in a decision (profiling only)"""
    def deco(f):
        def wrapped(self, *a, **kw):
            with self.at(role=name):
                return f(self, *a, **kw)
        wrapped.__name__, wrapped.__doc__ = f.__name__, f.__doc__
        return wrapped
    return deco


class Line(str):
    """an emitted instruction that remembers which loop section / role it belongs to (tools/emu/profile.py --sections)"""
    sec = ""
    role = ""


class QE(tuple):
    """a queued (deferred) instruction: (text, reads_sym, reads_vcc); `tag` = (section, role) of where it was queued (not part of
    the queue's identity: two paths that queue the same update arrive with the same queue)"""
    tag = ("", "")


class Gen:
    def __init__(self, lp0, pb4=False, hbm=False):
        set_layout()
        # hbm: lc + lp > 4 (legal in a .lzma header, lzma.rs:96-161; up to 4096 literal rows of 1536 bytes: 256 plain probabilities +
        # 2 x 256 matched ones).  The rows live in a slab in HBM (lit_rsrc: row r at r * 1536 -- 512 bytes plain, packed as in the
        # registers, then 1024 bytes matched, as in LDS); the eight row registers and the eight LDS rows become direct-mapped caches
        # (slot = row & 7) with their tags in the lanes of vtag / vtagm (-1: empty).  A row comes in when a literal needs it and the
        # slot holds another (one load + wait: the price of a miss), the evicted one goes back with a store nobody waits for.  The
        # row the literal walk is on (u0..u3) owns its slot.  Needs LITSPLIT's layout (gpr index = slot).
        self.hbm = hbm
        assert not hbm or LITSPLIT
        assert not hbm or (pb4 and not lp0) or (lp0 and not pb4)   # HBM: any lp / pb; HB0 (round 6): lp == 0, pb <= 2
        self.lp0 = lp0  # generate for lp == 0 (literal row = prev >> (8 - lc))
        # pb4: up to 16 position states.  is_match / is_rep0long [state * 16 + pos_state] span three registers (lanes
        # 0..63 / 64..127 / 128..191 by the index's bits 6-7), len low / mid [pos_state] two (pos_state bit 3; roots at lanes
        # 8 + (pos_state & 7), heap-numbered down to lane 63), the four len choices move to lanes 48..51 of m_rep.
        self.pb4 = pb4
        self.main, self.cold, self.cold2, self.stubs = [], [], [], []  # (cold2: out-of-line code of code that is itself in `cold`)
        self.cur = self.main
        self.uid = 0
        # deferred vector instructions (see DEFER): a FIFO of (text, reads_sym, reads_vcc); `reach`: the previous instruction can
        # fall through to the next one; lstate: label -> the queue every path must arrive with (checked at each branch / label)
        self.q = []
        self.split = LENDEFER and not pb4 and "tree" in DEFER and "tree" in FORMB
        self.lazy = self.split and ALIGNLAZY
        self.sq = []          # scalar (or index-mode) instructions for the next decision's shadow (SSHADOW): straight-line code only
        self.reach = True
        self.no_align = False   # (inside the direct-bit chains: ALIGN_FREE must not touch their layout)
        self.lstate = {}
        self.sec, self.role = "prologue", "book"   # attribution of what is emitted (profiling only)

    ANY_STATE = ("Xeof", "Xmatch_dist_dict", "Xmatch_dist_out")   # decoding ends there with an error: the model no longer matters

    def _st(self):
        return tuple(self.q)

    def _edge(self, name, conditional):
        if name in self.ANY_STATE:
            return
        reg = self.lstate.get(name)
        if reg is None:
            self.lstate[name] = self._st()
        elif reg != self._st():
            if not conditional and reg == ():
                self.flush()
            else:
                raise AssertionError("branch to %s with deferred updates %r, the label expects %r" % (name, self._st(), reg))

    def e(self, fmt, **kw):
        """emit one instruction; {x} is replaced by the register of operand/temporary x"""
        out = fmt
        for n in set(re.findall(r"\{(\w+)\}", fmt)):
            out = out.replace("{%s}" % n, kw[n] if n in kw else R(n))
        ops = out.replace(",", " ").split()
        op = ops[0]
        if op in ("s_branch", "s_call_b64", "s_setpc_b64") or op.startswith("s_cbranch"):
            assert not self.sq or kw.get("_normtest"), "a branch while scalar instructions wait for a shadow: " + out
            tgt = re.search(r"L(\w+)%=", out)
            if tgt and op != "s_call_b64":
                self._edge(tgt.group(1), op != "s_branch")
        elif len(ops) > 1 and not kw.get("_queued"):
            # a queued instruction must be emitted before something it reads is overwritten
            if S["sym"] == "m0" and (op == "s_set_gpr_idx_on" or (ops[1] == "m0" and op == "s_mov_b32")) and any(x[1] for x in self.q):
                raise AssertionError("m0 (the symbol) is overwritten while a deferred update still reads it: " + out)
            if ops[1] == S["sym"] and not op.startswith(("s_cmp", "s_bitcmp")) and any(x[1] for x in self.q):   # (compares only read it)
                raise AssertionError("sym is overwritten while a deferred update still reads it: " + out)
            if ops[1] == "vcc" and op.startswith("v_") and any(x[2] for x in self.q):
                raise AssertionError("vcc is overwritten while a deferred update still reads it: " + out)
        line = Line("  " + out)
        line.sec, line.role = kw.get("_tag") or (self.sec, self.role)
        self.cur.append(line)
        if op in ("s_branch", "s_setpc_b64"):
            self.reach = False

    def lab(self, name):
        assert not self.sq, "label %s while scalar instructions wait for a shadow" % name
        reg = self.lstate.get(name)
        if self.reach:
            if reg is None:
                self.lstate[name] = self._st()
            elif reg != self._st():
                if reg == ():
                    self.flush()
                else:
                    raise AssertionError("falling into %s with deferred updates %r, the label expects %r" % (name, self._st(), reg))
        else:
            if reg is None:
                reg = self.lstate[name] = ()
            self.q = list(reg)
        if ALIGN_FREE and not self.reach and not self.no_align and not name.startswith(("dchain", "dend", "dtr_", "db_", "dn_", "DN")):
            self.cur.append(".p2align %d" % ALIGN_FREE)
        self.reach = True
        self.cur.append("L%s%%=:" % name)

    @staticmethod
    def L(name):
        return "L%s%%=" % name

    def new(self, prefix):
        self.uid += 1
        return "%s%d" % (prefix, self.uid)

    class _Into:
        """emit into another list (out-of-line code): starts unreachable with an empty queue (its first label sets the state)"""
        def __init__(self, g, lst, keep=False):
            self.g, self.lst, self.keep = g, lst, keep

        def __enter__(self):
            self.saved = (self.g.cur, self.g.q, self.g.reach)
            self.g.cur = self.lst
            if not self.keep:
                self.g.q, self.g.reach = [], False

        def __exit__(self, *a):
            if (not a or a[0] is None) and not self.keep:
                assert not self.g.reach, "out-of-line code falls off its end"
            self.g.cur, self.g.q, self.g.reach = self.saved

    # ---- deferred vector instructions ----------------------------------------------------------------------------
    def defer(self, fmt, reads_sym=False, reads_vcc=False, **kw):
        out = fmt
        for n in set(re.findall(r"\{(\w+)\}", fmt)):
            out = out.replace("{%s}" % n, kw[n] if n in kw else R(n))
        q = QE((out, reads_sym, reads_vcc))
        q.tag = (self.sec, "update")
        self.q.append(q)

    def shadow(self, n):
        """up to n queued instructions, here"""
        k = 0
        while self.q and k < n:
            q = self.q.pop(0)
            self.e(q[0], _queued=True, _tag=getattr(q, "tag", None))
            k += 1
        return k

    def queue_s(self, fmt, kind="", **kw):
        """a scalar instruction nothing waits for: into the next decision's shadow (SSHADOW lists the kinds), else here"""
        if kind not in SSHADOW:
            return self.e(fmt, **kw)
        self.sq.append((fmt, kw, (self.sec, self.role)))

    def shadow_s(self):
        sq, self.sq = self.sq, []
        for fmt, kw, tag in sq:
            self.e(fmt, _tag=tag, **kw)

    def flush(self):
        while self.q:
            q = self.q.pop(0)
            self.e(q[0], _queued=True, _tag=getattr(q, "tag", None))

    def flush_reads(self, sym=False, vcc=False):
        """everything up to the last queued instruction that reads sym / vcc"""
        last = max([i for i, x in enumerate(self.q) if (sym and x[1]) or (vcc and x[2])], default=-1)
        for _ in range(last + 1):
            q = self.q.pop(0)
            self.e(q[0], _queued=True, _tag=getattr(q, "tag", None))

    def in_cold(self):
        return Gen._Into(self, self.cold)

    class _At:
        """with g.at(sec=..., role=...): what is emitted inside is attributed to that section / role (profiling only)"""
        def __init__(self, g, sec, role):
            self.g, self.new = g, (sec, role)

        def __enter__(self):
            self.saved = (self.g.sec, self.g.role)
            if self.new[0] is not None:
                self.g.sec = self.new[0]
            if self.new[1] is not None:
                self.g.role = self.new[1]

        def __exit__(self, *a):
            self.g.sec, self.g.role = self.saved

    def at(self, sec=None, role=None):
        return Gen._At(self, sec, role)

    # ---- range coder ------------------------------------------------------------------------------
    def norm(self, to=None, kind="single"):
        """RangeDecoder::normalize (rangecoder.rs:59-69) as a check + out-of-line stub.
        `to`: label to continue at (default: fall through)."""
        k = self.new("N")
        if to is not None and self.q and self.lstate.get(to, ()) == ():
            self.flush()          # (the stub branches to `to` as well: both must arrive with what the label expects)
        with self.at(role="normtest"):
            if kind in NORM_S:
                self.e("s_cmp_lt_u32 {range}, " + ("{pad}" if K24S else "0x1000000"))
                self.e("s_cbranch_scc1 " + self.L(k))
            else:
                self.e("v_cmp_lt_u32 vcc, {range}, {VKTOP}")     # all lanes agree
                self.e("s_cbranch_vccnz " + self.L(k))
        if to is None:
            ret = self.new("R")
            self.lab(ret)
        else:
            ret = to
            with self.at(role="book"):
                self.e("s_branch " + self.L(to))
        with Gen._Into(self, self.stubs), self.at(role="normstub"):
            if ALIGN_STUBS:
                self.cur.append(".p2align %d" % ALIGN_STUBS)
            self.lab(k)
            if not EOFWRAP:
                self.e("s_cmp_eq_u32 {off}, {lim}")
                self.e("s_cbranch_scc1 " + self.L("Xeof"))
            if not NBPRE:
                self.e("v_readlane_b32 {n1}, {winb}, {off}")
            if NORM64:   # range < 2^24 here: its top byte is zero, so one 64-bit shift of the (range, code) pair moves both
                self.e("s_lshl_b64 " + RC + ", " + RC + ", 8")
            else:
                self.e("s_lshl_b32 {range}, {range}, 8")
                self.e("s_lshl_b32 {code}, {code}, 8")
            self.e("s_or_b32 {code}, {code}, " + ("{nb}" if NBPRE else "{n1}"))
            self.e("s_add_u32 {off}, {off}, 1")
            if NBPRE:    # the byte after it, for the next normalisation (a used-up window: garbage, the refill reads again)
                self.e("v_readlane_b32 {nb}, {winb}, {off}")
            if not OFFBIAS:
                self.e("s_bitcmp1_b32 {off}, 6")
            self.e("s_cbranch_scc0 " + self.L(ret))
            self.e("s_call_b64 " + RET + ", " + self.L("refill"))
            self.e("s_branch " + self.L(ret))

    def pad(self):
        for _ in range(PAD_S):
            self.e("s_mov_b32 {pad}, 0")
        for _ in range(PAD_V):
            self.e("v_mov_b32 {vpad}, 0")
        for _ in range(PAD_B):
            self.e("s_cbranch_execz " + self.L("Xeof"))     # (never taken; Xeof accepts any queue state)

    @role("core")
    def core(self, T, ln, half=None, cmp_lane=None, formb=False):
        """decode_bit (rangecoder.rs:92-120) on the probability in lane `ln` of T, up to the point where
        SCC = (bit == 0) and range / code are updated.  half: None = T holds one probability per lane;
        0 / 1 = low / high 16 bits (then vx = this lane's probability).  vcc = mask of lane `ln`.
        Queued vector instructions (DEFER) go into the shadow of the v_readlane(s); those that read the previous decision's
        vcc all go before this decision's own v_cmp_eq."""
        self.pad()
        mask = lambda: self.e("v_cmp_eq_u32 vcc, {ln}, {cl}", ln=ln, cl=cmp_lane or R("v_lane"))
        if ("lit" if half is not None else "single") in FORMB or formb:
            rs = ("lit" if half is not None else "single") in R11S
            self.e("s_lshr_b32 {st}, {range}, 11" if rs else "v_lshrrev_b32 {vt}, 11, {range}")
            if half == 0:
                self.e("v_and_b32 {vx}, 0xffff, {T}", T=T)
            elif half == 1:
                self.e("v_lshrrev_b32 {vx}, 16, {T}", T=T)
            self.e("v_mul_u32_u24 {vb}, {t}, {src}", t=R("st") if rs else R("vt"), src=T if half is None else R("vx"))
            self.e("v_sub_u32 {vr}, {range}, {vb}")
            self.e("v_readlane_b32 {range}, {vb}, {ln}", ln=ln)      # (bound, code)
            self.e("v_readlane_b32 {sr1}, {vr}, {ln}", ln=ln)        # range - bound
            self.shadow(SHADOW)
            self.flush_reads(vcc=True, sym=True)     # (the caller may extend sym right after the decision)
            self.shadow_s()
            mask()
            self.e("s_sub_u32 {sc1}, {code}, {range}")               # SCC = code < bound  <=>  bit == 0
            self.e("s_cselect_b64 " + RC + ", " + RC + ", " + RC1)
            return
        if half is None and "single" in S1:
            # all scalar: the node's probability itself is read out, the bound is s_mul_i32's
            self.e("v_readlane_b32 {n1}, {T}, {ln}", T=T, ln=ln)
            self.shadow(SHADOW)
            self.flush_reads(vcc=True, sym=True)
            self.shadow_s()
            mask()
            self.e("s_lshr_b32 {st}, {range}, 11")
            self.e("s_mul_i32 {sb}, {st}, {n1}")
            self.e("s_sub_u32 {sr1}, {range}, {sb}")
            self.e("s_sub_u32 {sc1}, {code}, {sb}")          # SCC = code < bound  <=>  bit == 0
            self.e("s_cselect_b32 {range}, {sb}, {sr1}")
            self.e("s_cselect_b32 {code}, {code}, {sc1}")
            return
        # form A: every lane computes the bound of its own probability; the one that is needed is read out
        rs = ("lit" if half is not None else "single") in R11S
        self.e("s_lshr_b32 {st}, {range}, 11" if rs else "v_lshrrev_b32 {vt}, 11, {range}")
        if half == 0:
            self.e("v_and_b32 {vx}, 0xffff, {T}", T=T)
        elif half == 1:
            self.e("v_lshrrev_b32 {vx}, 16, {T}", T=T)
        self.e("v_mul_u32_u24 {vb}, {t}, {src}", t=R("st") if rs else R("vt"), src=T if half is None else R("vx"))
        if self.q:                # (one instruction between vb's producer and the v_readlane of it: a queued one, else the mask)
            self.shadow(1)
            self.e("v_readlane_b32 {sb}, {vb}, {ln}", ln=ln)
            self.shadow(SHADOW - 1)
            self.flush_reads(vcc=True, sym=True)
            self.shadow_s()
            mask()
        else:
            mask()
            self.e("v_readlane_b32 {sb}, {vb}, {ln}", ln=ln)
            self.shadow_s()
        self.e("s_sub_u32 {sr1}, {range}, {sb}")
        self.e("s_sub_u32 {sc1}, {code}, {sb}")          # SCC = code < bound  <=>  bit == 0
        self.e("s_cselect_b32 {range}, {sb}, {sr1}")
        self.e("s_cselect_b32 {code}, {code}, {sc1}")

    @role("update")
    def _apply(self, T, half):
        self.e("v_ashrrev_i32 {vt}, 5, {vt}")            # (K - p) >> 5, arithmetic
        if half == 1:
            self.e("v_lshlrev_b32 {vt}, 16, {vt}")
        self.e("v_add_u32 {vt}, {T}, {vt}", T=T)        # low half: the sign extension of a negative delta cancels
        self.e("v_cndmask_b32 {T}, {T}, {vt}, vcc", T=T)  # against the carry out of the (non-negative) low half

    # p += (K - p) >> 5 with K = 2048 for a 0 bit and 31 for a 1 bit (the arithmetic shift makes the second
    # -(p >> 5)); for an unpacked probability this is (31 * p + K) >> 5: one v_mad and one shift.
    @role("update")
    def post_known(self, T, bit0, half=None, defer=True):
        """probability update when the bit value is known from the branch taken"""
        if half is None:
            k = "{c2048}" if bit0 else "31"
            if defer and "single" in DEFER:      # into the next decision's shadow (its v_cmp_eq comes after these)
                self.defer("v_mad_u32_u24 {DVT}, {T}, 31, %s" % k, T=T)
                self.defer("v_lshrrev_b32 {DVT}, 5, {DVT}")
                self.defer("v_cndmask_b32 {T}, {T}, {DVT}, vcc", reads_vcc=True, T=T)
                return
            self.e("v_mad_u32_u24 {vt}, {T}, 31, %s" % k, T=T)
            self.e("v_lshrrev_b32 {vt}, 5, {vt}")
            self.e("v_cndmask_b32 {T}, {T}, {vt}, vcc", T=T)
            return
        self.e("v_sub_u32 {vt}, %s, {vx}" % ("0x800" if bit0 else "31"))
        self._apply(T, half)

    @role("update")
    def pre_sym(self):
        """scalar part of the update of a tree decision; call while SCC = (bit == 0)"""
        if not PRESYMV:
            self.e("s_cselect_b32 {sk}, 0x800, 31")

    @role("update")
    def post_sym(self, T, half=None, defer=True):
        """probability update of a tree decision (the symbol's new low bit is 1 if the bit was 0)"""
        if PRESYMV and half is None and defer and "single" in DEFER:
            # (the symbol's low bit is 1 for a 0 bit: K = 31 + 2017 * (sym & 1); read before the next decision extends the symbol)
            self.defer("v_and_b32 {DVX}, 1, {sym}", reads_sym=True)
            self.defer("v_mad_u32_u24 {DVX}, {DVX}, {c2017}, 31")
            self.defer("v_mad_u32_u24 {DVT}, {T}, 31, {DVX}", T=T)
            self.defer("v_lshrrev_b32 {DVT}, 5, {DVT}")
            self.defer("v_cndmask_b32 {T}, {T}, {DVT}, vcc", reads_vcc=True, T=T)
            return
        if PRESYMV:
            self.e("s_and_b32 {sk}, {sym}, 1")
            self.e("s_mul_i32 {sk}, {sk}, 2017")
            self.e("s_add_u32 {sk}, {sk}, 31")
        if half is None:
            if defer and "single" in DEFER:      # (sk stays valid: the next decision's pre_sym comes after its shadow)
                self.defer("v_mad_u32_u24 {DVT}, {T}, 31, {sk}", T=T)
                self.defer("v_lshrrev_b32 {DVT}, 5, {DVT}")
                self.defer("v_cndmask_b32 {T}, {T}, {DVT}, vcc", reads_vcc=True, T=T)
                return
            self.e("v_mad_u32_u24 {vt}, {T}, 31, {sk}", T=T)
            self.e("v_lshrrev_b32 {vt}, 5, {vt}")
            self.e("v_cndmask_b32 {T}, {T}, {vt}, vcc", T=T)
            return
        self.e("v_sub_u32 {vt}, {sk}, {vx}")
        self._apply(T, half)

    @role("core")
    def bit(self, T, ln, half=None, first=False, cmp_lane=None, defer=True):
        """one tree decision: sym = 2 * sym + (bit == 0), then normalise.  first: sym was 1 (not
        materialised), ln is the constant lane of the root."""
        acc = "s_cselect_b32 {sym}, 3, 2" if first else "s_addc_u32 {sym}, {sym}, {sym}"
        self.core(T, ln, half, cmp_lane)
        self.pre_sym()
        self.e(acc)
        self.post_sym(T, half, defer=defer)
        self.norm()

    @role("core")
    def bit_nu(self, T, ln, first=False):
        """one decision of a tree whose probabilities are updated after the walk (tree_update): 5 scalar +
        4 vector instructions and a wait state"""
        e = self.e
        self.pad()
        if "tree" in FORMB:
            if "tree" in R11S:
                e("s_lshr_b32 {st}, {range}, 11")
                e("v_mul_u32_u24 {vb}, {st}, {T}", T=T)
            else:
                e("v_lshrrev_b32 {vt}, 11, {range}")
                e("v_mul_u32_u24 {vb}, {vt}, {T}", T=T)
            e("v_sub_u32 {vr}, {range}, {vb}")
            e("v_readlane_b32 {range}, {vb}, {ln}", ln=ln)           # (bound, code)
            e("v_readlane_b32 {sr1}, {vr}, {ln}", ln=ln)             # range - bound
            self.shadow(SHADOW)
            self.flush_reads(sym=True)                               # (sym changes below)
            self.shadow_s()
            e("s_sub_u32 {sc1}, {code}, {range}")                    # SCC = code < bound  <=>  bit == 0
            e("s_cselect_b64 " + RC + ", " + RC + ", " + RC1)
            e("s_cselect_b32 {sym}, 3, 2" if first else "s_addc_u32 {sym}, {sym}, {sym}")
            self.norm(kind="tree")
            return
        if FORMA2:
            e("v_lshrrev_b32 {vt}, 11, {range}")
            e("v_mul_u32_u24 {vb}, {vt}, {T}", T=T)
            if not self.shadow(1):                                   # (one instruction between vb's producer and the v_readlane of it)
                e("s_nop 0")
            e("v_readlane_b32 {sb}, {vb}, {ln}", ln=ln)
            self.shadow(SHADOW - 1)
            self.flush_reads(sym=True)
            self.shadow_s()
            e("s_sub_u32 {sr1}, {range}, {sb}")
            e("s_sub_u32 {sc1}, {code}, {sb}")                       # SCC = code < bound  <=>  bit == 0
            e("s_cselect_b32 {range}, {sb}, {sr1}")
            e("s_cselect_b32 {code}, {code}, {sc1}")
            e("s_cselect_b32 {sym}, 3, 2" if first else "s_addc_u32 {sym}, {sym}, {sym}")
            self.norm(kind="tree")
            return
        self.flush()
        self.shadow_s()
        e("v_lshrrev_b32 {vt}, 11, {range}")
        e("v_mul_u32_u24 {vb}, {vt}, {T}", T=T)
        e("s_nop 0")  # gfx940: one wait state between a VALU write and the v_readlane of it (measured: without it every stream decodes wrongly)
        e("v_readlane_b32 {sb}, {vb}, {ln}", ln=ln)
        e("s_sub_u32 {sr1}, {range}, {sb}")
        e("s_sub_u32 {sc1}, {code}, {sb}")                 # SCC = code < bound  <=>  bit == 0
        e("s_cselect_b32 {range}, {sb}, {sr1}")
        e("s_cselect_b32 {code}, {code}, {sc1}")
        e("s_cselect_b32 {sym}, 3, 2" if first else "s_addc_u32 {sym}, {sym}, {sym}")
        self.norm(kind="tree")

    def tree_walk(self, T, nbits, first_lane=None):
        """nbits decisions down a heap-numbered tree in T (lane = running symbol).  first_lane: constant
        lane of the root when the symbol starts at 1 (then it is not materialised beforehand)."""
        for i in range(nbits):
            first = first_lane is not None and i == 0
            self.bit_nu(T, first_lane if first else R("sym"), first=first)

    @role("update")
    def tree_update(self, T, final_level, min_level=None, defer=False):
        """The probability updates of a walked tree, all at once: the final symbol (at heap level
        `final_level`, inverted-bit path) names the visited node of every level (its prefixes) and the bit
        decided there (the next bit down); lane L was visited iff sym >> (final_level - level(L)) == L.
        Every visited lane becomes (31 p + K) >> 5, K = 2048 if the bit was 0 (inverted bit 1) else 31.
        min_level: SGPR; only nodes at that heap level or below it were walked in this table."""
        e = self.e
        sh, shm1 = {6: ("VSH6", "VSH6M1"), 5: ("VSH5", "VSH5M1"), 4: ("VSH4", "VSH4M1")}[final_level]
        if defer and min_level is None and "tree" in DEFER:
            # queued for the shadows of the decisions that follow: own temporaries (DVA, DVX) and mask (DM), the instructions
            # that read sym first
            d = self.defer
            d("v_lshrrev_b32 {DVA}, {sh}, {sym}", reads_sym=True, sh=R(sh))
            d("v_bfe_u32 {DVX}, {sym}, {sh}, 1", reads_sym=True, sh=R(shm1))
            d("v_cmp_eq_u32_e64 " + DM + ", {DVA}, {v_lane}")
            d("v_mad_u32_u24 {DVX}, {DVX}, {c2017}, 31")
            d("v_mad_u32_u24 {DVA}, {T}, 31, {DVX}", T=T)
            d("v_lshrrev_b32 {DVA}, 5, {DVA}")
            d("v_cndmask_b32_e64 {T}, {T}, {DVA}, " + DM, T=T)
            return
        self.flush()
        e("v_lshrrev_b32 {va}, {sh}, {sym}", sh=R(sh))
        e("v_bfe_u32 {vx}, {sym}, {sh}, 1", sh=R(shm1))
        e("v_cmp_eq_u32 vcc, {va}, {v_lane}")
        e("v_mad_u32_u24 {vx}, {vx}, {c2017}, 31")
        if min_level is not None:
            e("v_cmp_le_u32 " + MPAIR + ", {m}, {VLEVEL}", m=min_level)
        e("v_mad_u32_u24 {vt}, {T}, 31, {vx}", T=T)
        e("v_lshrrev_b32 {vt}, 5, {vt}")
        if min_level is not None:
            e("s_and_b64 vcc, vcc, " + MPAIR)
        e("v_cndmask_b32 {T}, {T}, {vt}, vcc", T=T)

    @role("update")
    def align_pending(self, queued):
        """the align tree's update from asym (ALIGNLAZY); queued: into the coming shadows, else here"""
        put = self.defer if queued else (lambda fmt, **kw: self.e(fmt, **{k: v for k, v in kw.items() if k not in ("reads_sym", "reads_vcc")}))
        T = R("m_align")
        put("v_lshrrev_b32 {DVA}, {sh}, {asym}", sh=R("VSH4"))
        put("v_bfe_u32 {DVX}, {asym}, {sh}, 1", sh=R("VSH4M1"))
        put("v_cmp_eq_u32_e64 " + DM + ", {DVA}, {v_lane}")
        put("v_mad_u32_u24 {DVX}, {DVX}, {c2017}, 31")
        put("v_mad_u32_u24 {DVA}, {T}, 31, {DVX}", T=T)
        put("v_lshrrev_b32 {DVA}, 5, {DVA}")
        put("v_cndmask_b32_e64 {T}, {T}, {DVA}, " + DM, T=T)
        put("s_mov_b32 {asym}, 0")

    @role("core")
    def decide(self, T, ln, taken, cmp_lane=None, defer=True):
        """a decision that ends in a branch: falls through for a 0 bit, jumps to `taken` for a 1 bit.
        The code at `taken` must start with self.taken(T)."""
        self.core(T, ln, cmp_lane=cmp_lane)
        self.e("s_cbranch_scc0 " + self.L(taken))
        self.post_known(T, True, defer=defer)
        self.norm()

    def decide_by_reg(self, regs, ln, taken_pfx, join):
        """decide() on probability `ln` (0..191) of a table that spans three registers; falls through at `join` for a 0
        bit; a 1 bit arrives at taken_pfx + "2" with the update done (each register has its own taken stub)."""
        e, L = self.e, self.L
        e("s_cmp_lt_u32 {ln}, 64", ln=ln)
        e("s_cbranch_scc0 " + L(taken_pfx + "_hi"))
        self.decide(R(regs[0]), ln, taken_pfx + "_ta", defer=False)   # (v_readlane takes the index's low 6 bits; the update mask compares
        self.lab(join)                                    #  it with lane + 64 / lane + 128 for the other two registers)
        with self.in_cold():
            self.lab(taken_pfx + "_hi")
            e("s_cmp_lt_u32 {ln}, 0x80", ln=ln)
            e("s_cbranch_scc0 " + L(taken_pfx + "_hi2"))
            self.decide(R(regs[1]), ln, taken_pfx + "_tb", cmp_lane=V["VLANE64"], defer=False)   # (the three registers' paths
            e("s_branch " + L(join))                                                             #  join: nothing queued there)
            self.lab(taken_pfx + "_hi2")
            self.decide(R(regs[2]), ln, taken_pfx + "_tc", cmp_lane=V["VLANE128"], defer=False)
            e("s_branch " + L(join))
            for tag, reg in zip("abc", regs):
                self.lab(taken_pfx + "_t" + tag)
                self.taken(R(reg), to=taken_pfx + "2", defer=False)

    def taken(self, T, to=None, defer=True):
        self.post_known(T, False, defer=defer)
        self.norm(to)

    @role("core")
    def direct_bit(self, acc, test=True, vec=False):
        """RangeDecoder::get_bit (rangecoder.rs:71-82): acc = 2 * acc + (bit == 0).  test=False: the caller knows where the
        normalisations fall (DIRECT8)."""
        if vec:
            assert not test
            self.e("v_lshrrev_b32 {vr}, 1, {vr}")
            self.e("v_addc_co_u32 {va}, vcc, {va}, {va}, vcc")     # (the bit before this one; entered with vcc = 0)
            self.e("v_sub_co_u32 {vx}, vcc, {vb}, {vr}")           # vcc = code < range
            self.e("v_min_u32 {vb}, {vb}, {vx}")                   # code - range wraps above code exactly when code < range
            return
        self.e("s_lshr_b32 {range}, {range}, 1")
        self.e("s_sub_u32 {sc1}, {code}, {range}")
        self.e("s_cselect_b32 {code}, {code}, {sc1}")
        self.e("s_addc_u32 {a}, {a}, {a}", a=acc)
        if test:
            self.norm(kind="direct")

    @role("normstub")
    def direct_norm(self, mark=None, vec=False, slot=None):
        """RangeDecoder::normalize (rangecoder.rs:59-69), unconditional and inline: range < 2^24 is known here"""
        e, L = self.e, self.L
        k = self.new("DN")
        if mark:
            self.lab(mark + "_s")
        if not EOFWRAP:
            e("s_cmp_eq_u32 {off}, {lim}")
            e("s_cbranch_scc1 " + L("Xeof"))
        if not NBPRE:
            e("v_readlane_b32 {n1}, {winb}, {off}")
        if vec and not VDIRECT_S:
            e("v_lshlrev_b32 {vr}, 8, {vr}")
            e("v_lshl_or_b32 {vb}, {vb}, 8, {nb}")
        elif vec:                                # (chains of both forms: every normalisation block the same size AND the same number of
            e("v_lshlrev_b32 {vr}, 8, {vr}")     #  instructions -- the emulator's addresses count instructions --: 28 bytes, six instructions)
            e("v_lshlrev_b32 {vb}, 8, {vb}")
            e("v_or_b32 {vb}, {nb}, {vb}")
        else:
            if VDIRECT and VDIRECT_S:
                e("s_nop 0")
            e("s_lshl_b64 " + RC + ", " + RC + ", 8")
            e("s_or_b32 {code}, {code}, " + ("{nb}" if NBPRE else "{n1}"))
        e("s_add_u32 {off}, {off}, 1")
        if NBPRE:
            e("v_readlane_b32 {nb}, {winb}, {off}")
        if not OFFBIAS:
            e("s_bitcmp1_b32 {off}, 6")
        e("s_cbranch_scc1 " + L(k))
        self.lab(k + "r")
        if slot:                                 # (QDIRECT: one more instruction per normalisation block -- the way to the quotient block, or s_nop)
            e(slot)
        if mark and not self.reach:
            self.cur.append("L%s%%=:" % (mark + "_e"))   # (behind an unconditional branch: a mark for the block's size, nothing arrives here)
        elif mark:
            self.lab(mark + "_e")
        with Gen._Into(self, self.stubs):
            self.lab(k)
            e("s_call_b64 " + RET + ", " + L("refill"))
            e("s_branch " + L(k + "r"))

    @role("core")
    def direct_quotient(self, name, j, cont, serial):
        """QDIRECT: j direct bits at once, out of line; cont: where the chain goes on behind them, serial: the bit block that does them one by one"""
        e, L = self.e, self.L
        m = (1 << j) - 1
        with Gen._Into(self, self.stubs):
            self.lab(name)
            e("s_lshr_b32 {range}, {range}, %d" % j)
            e("v_xor_b32 {vx}, %d, {v_lane}" % m)
            if j < 6:
                e("v_and_b32 {vx}, %d, {vx}" % m)
            e("v_mul_lo_u32 {vx}, {vx}, {range}")
            e("v_sub_u32 {vx}, {code}, {vx}")
            e("v_cmp_gt_u32 vcc, {range}, {vx}")
            e("s_cbranch_vccz " + L(name + "f"))
            e("s_ff1_i32_b64 {t0}, vcc")                 # the inverted quotient
            e("s_xor_b32 {t1}, {t0}, %d" % m)
            e("s_mul_i32 {t1}, {t1}, {range}")
            e("s_sub_u32 {code}, {code}, {t1}")
            e("s_lshl_b32 {t4}, {t4}, %d" % j)
            e("s_or_b32 {t4}, {t4}, {t0}")
            e("s_branch " + L(cont))
            self.lab(name + "f")                         # code >= range: bit by bit, as get_bit has it
            e("s_lshl_b32 {range}, {range}, %d" % j)
            e("s_branch " + L(serial))

    @role("book")
    def direct_cross(self):
        """VDIRECT: from the chain's vector registers to (range, code, t4) -- with the last bit's borrow, which the next bit block would
        have added (two instructions between the v_sub_co that wrote vcc and its reader, one between a VALU write and its v_readfirstlane)"""
        e = self.e
        e("v_readfirstlane_b32 {range}, {vr}")
        e("v_addc_co_u32 {va}, vcc, {va}, {va}, vcc")
        e("v_readfirstlane_b32 {code}, {vb}")
        e("v_readfirstlane_b32 {t4}, {va}")

    # ---- pending short match ---------------------------------------------------------------------------
    def finish_pending(self, have_t6=False, prof=None, extract=True):
        if WAITPROF and prof:
            self.e("s_memtime s[94:95]")
            self.e("s_waitcnt lgkmcnt(0)")
            self.e("s_mov_b32 s96, s94")
            self.e("s_waitcnt vmcnt(0)")
            self.e("s_memtime s[94:95]")
            self.e("s_waitcnt lgkmcnt(0)")
            self.e("s_sub_u32 s96, s94, s96")
            self.e("s_add_u32 {w}, {w}, s96", w=R("prof_w" + prof))
            self.e("s_add_u32 {n}, {n}, 1", n=R("prof_n" + prof))
        # (gfx940 family: a VALU write of VCC / an SGPR wants 2 wait states before a VALU read of it, hence
        #  the order: the v_cmp that writes vcc is never followed directly by the v_cndmask that reads it)
        self.e("s_waitcnt vmcnt(0)")
        self.e("v_cmp_gt_u32_e64 " + MPAIR + ", {pend_n}, {v_lane}")   # (not vcc: a deferred update may still read it)
        self.e("v_add_u32 {VT0}, {pend_pos}, {v_lane}")
        if extract:
            if not have_t6:
                self.e("s_add_u32 {t6}, {pend_n}, -1")
            self.e("v_readlane_b32 {prev}, {pend_val}, {t6}")
            self.e("v_readlane_b32 {mb}, {pend_val}, {pend_n}")
        else:                                              # (keeps the v_cmp two instructions away from the v_cndmask)
            self.e("s_nop 0")
        self.e("v_cndmask_b32_e64 {VT0}, -1, {VT0}, " + MPAIR)
        self.e("buffer_store_byte {pend_val}, {VT0}, {out_rsrc}, 0 offen" + STORE_MOD)
        if extract:                                        # (extract=False: a match follows and sets pend_n itself)
            self.e("s_mov_b32 {pend_n}, 0")

    def set_guards(self, t):
        """gtop / gdist from len, lim, target, safe_len, mlen (prologue and window refill; clobbers SCC and t)"""
        e = self.e
        e("s_add_u32 {t}, {safe_len}, 1", t=t)
        e("s_min_u32 {gtop}, {target}, {t}", t=t)
        e("s_min_u32 {gtop}, {gtop}, {qtop}")             # the scheduler's quantum: yield at the first symbol top with len >= qtop
        # gtop = 0 (every symbol looks closer) while the reader may be at EOF before the symbol ends.  (lim is a u32 of up to 4 GiB:
        # no signed compares on it)
        if EOFWRAP:
            # every symbol looks closer once the reader is at EOF (lim = -1) -- with FEED: from the view's last window on (lim = 0, -1)
            if FEED:
                assert FEED_GUARD >= FEED_MARGIN
                e("s_and_b32 {gdist}, {lc8}, %d" % FEED_GUARD)     # (gdist: set below) 0, or FEED_GUARD when the view will be continued
                e("s_add_u32 {t}, {lim}, 1", t=t)
                e("s_cmp_gt_u32 {t}, {gdist}", t=t)               # lim >= FEED_GUARD: no symbol that starts in this window can reach the view's end
            else:
                e("s_cmp_lg_u32 {lim}, -1")                     # -1: the reader IS at EOF
        elif OFFBIAS:
            e("s_add_u32 {t}, {lim}, 64", t=t)
            e("s_cmpk_gt_u32 {t}, 63", t=t)
        else:
            e("s_cmpk_gt_u32 {lim}, 63")                    # more than this window left
        e("s_cselect_b32 {gtop}, {gtop}, 0")
        e("s_min_u32 {gdist}, {len}, {dict_size}")          # a lower bound of min(len, dict_size): len only grows
        e("s_cmpk_ge_u32 {mlen}, 64")
        e("s_cselect_b32 {gdist}, 0, {gdist}")
        e("s_cmp_gt_u32 {len}, {safe_len}")
        e("s_cselect_b32 {gdist}, 0, {gdist}")

    def tables_prologue(self):
        """tbl_a[lane] / tbl_b[lane] for lane = the pos_slot walk's final symbol & 63 (the inverted path: slot = lane ^ 63):
        tbl_b = byte offset from Lbase of the code that continues this slot's distance (slots >= 14: the entry into the
        chain of 26 direct-bit blocks that leaves exactly (slot >> 1) - 5 of them to run; 4..13: dist_rev; < 4:
        dist_small); tbl_a = ((3 + (slot & 1)) << ndb) - 1 for slots >= 14 (what the inverted direct / align bits are
        subtracted from, lzma.rs:576-590), (2 | (slot & 1)) << ndb for 4..13, the slot itself below 4.
        Computed once per unit and loop variant (tbl_ready), the base address once per entry."""
        e, L = self.e, self.L
        e("s_getpc_b64 " + JBASE)
        self.lab("base")
        vid = (5 if self.lp0 else 4) if self.hbm else 3 if self.pb4 else 1 if self.lp0 else 2  # (tbl_b holds offsets into THIS variant of the loop: an LZMA2
        e("s_cmp_eq_u32 {tbl_ready}, %d" % vid)             #  unit may change lp, and with it the variant, between chunks)
        e("s_cbranch_scc1 " + L("tbl_done"))
        bs = "(" + L("direct_done") + "-" + L("direct_chain") + ")/26"
        e("v_xor_b32 {VT0}, 63, {v_lane}")                 # slot
        e("v_lshrrev_b32 {VT1}, 1, {VT0}")
        e("v_and_b32 {VT2}, 1, {VT0}")
        if DIRECT8:
            # (offset << 8) | n.  Slots >= 14: n = (slot >> 1) - 5 direct bits, offset = chain 0's end minus n bit blocks; below 14:
            # n = 0 and the offset of chain 0's trampoline to dist_rev / dist_small (the jump adds clz chains: every chain has one)
            e("v_add_u32 {vb}, -5, {VT1}")                 # n
            e("v_mul_u32_u24 {vt}, (" + L("db_e") + "-" + L("db_s") + "), {vb}")
            e("v_sub_u32 {tbl_b}, " + L("dend0") + "-" + L("base") + ", {vt}")
            if VDIRECT and VDIRECT_S:                      # (an entry with more than VDIRECT_S bits to go lies in front of the chain's crossing block)
                e("v_mov_b32 {vx}, " + L("dt_e") + "-" + L("dt_s"))
                e("v_cmp_lt_u32 vcc, %d, {vb}" % VDIRECT_S)
                e("s_nop 0")
                e("s_nop 0")
                e("v_cndmask_b32 {vx}, 0, {vx}, vcc")
                e("v_sub_u32 {tbl_b}, {tbl_b}, {vx}")
            if DISP2:                                      # B2 = offset + n * chain stride (see distance_tables)
                e("v_mad_u32_u24 {tbl_b}, {vb}, {VCH}, {tbl_b}")
            e("v_lshl_or_b32 {tbl_b}, {tbl_b}, 8, {vb}")
        else:
            e("v_sub_u32 {vt}, 31, {VT1}")                     # 26 - count
            e("v_mul_u32_u24 {vt}, " + bs + ", {vt}")
            e("v_add_u32 {tbl_b}, " + L("direct_chain") + "-" + L("base") + ", {vt}")
        rev, small = ("dtr_rev0", "dtr_small0") if DIRECT8 else ("dist_rev", "dist_small")
        sh = "*256" if DIRECT8 else ""
        e("v_mov_b32 {vx}, (" + L(rev) + "-" + L("base") + ")" + sh)
        e("v_cmp_gt_u32 vcc, 14, {VT0}")
        e("s_nop 0")
        e("s_nop 0")
        e("v_cndmask_b32 {tbl_b}, {tbl_b}, {vx}, vcc")
        e("v_mov_b32 {vx}, (" + L(small) + "-" + L("base") + ")" + sh)
        e("v_cmp_gt_u32 vcc, 4, {VT0}")
        e("s_nop 0")
        e("s_nop 0")
        e("v_cndmask_b32 {tbl_b}, {tbl_b}, {vx}, vcc")
        e("v_add_u32 {vb}, -1, {VT1}")                     # ndb
        e("v_add_u32 {vt}, 3, {VT2}")
        e("v_lshlrev_b32 {vt}, {vb}, {vt}")
        e("v_add_u32 {vt}, -1, {vt}")
        e("v_add_u32 {vx}, 2, {VT2}")
        e("v_lshlrev_b32 {vx}, {vb}, {vx}")
        e("v_cmp_gt_u32 vcc, 14, {VT0}")
        e("s_nop 0")
        e("s_nop 0")
        e("v_cndmask_b32 {tbl_a}, {vt}, {vx}, vcc")
        e("v_cmp_gt_u32 vcc, 4, {VT0}")
        e("s_nop 0")
        e("s_nop 0")
        e("v_cndmask_b32 {tbl_a}, {tbl_a}, {VT0}, vcc")
        e("s_mov_b32 {tbl_ready}, %d" % vid)
        self.lab("tbl_done")

    def set_prio(self, reg, tmp):
        """s_setprio takes an immediate: a short chain picks the instruction for reg & 3 (clobbers SCC and tmp)"""
        done = self.new("P")
        self.e("s_and_b32 {t}, {r}, 3", r=reg, t=tmp)
        for k in range(3):
            self.e("s_setprio %d" % k)
            self.e("s_cmp_eq_u32 {t}, %d" % k, t=tmp)
            self.e("s_cbranch_scc1 " + self.L(done))
        self.e("s_setprio 3")
        self.lab(done)

    def prof_begin(self):
        self.e("s_waitcnt vmcnt(0)")                       # nothing older in flight: the next operation is timed alone
        self.e("s_memtime s[94:95]")
        self.e("s_waitcnt lgkmcnt(0)")
        self.e("s_mov_b32 s96, s94")

    def prof_end(self, which):
        self.e("s_waitcnt vmcnt(0)")
        self.e("s_memtime s[94:95]")
        self.e("s_waitcnt lgkmcnt(0)")
        self.e("s_sub_u32 s96, s94, s96")
        self.e("s_add_u32 {w}, {w}, s96", w=R("prof_w" + which))
        self.e("s_add_u32 {n}, {n}, 1", n=R("prof_n" + which))

    def exit_with(self, code):
        self.e("s_mov_b32 {exitcode}, %d" % EXIT[code])
        self.e("s_branch " + self.L("finish"))

    # ---- LenDecoder::decode ------------------------------------------------------------------------------
    def len_tree3(self, reg, reg_b, tag, defer=False):
        """the 3-bit tree low[pos_state] / mid[pos_state]; leaves the inverted path in sym's low 3 bits.
        4 position states: roots at lanes 4 + ps of `reg` (heap-numbered to lane 31).  16 (pb4): roots at lanes
        8 + (ps & 7) of `reg` / `reg_b` by ps bit 3 (heap-numbered to lane 63)."""
        e, L = self.e, self.L
        if not self.pb4:
            e("s_add_u32 {sym}, {ps}, 4")
            self.tree_walk(R(reg), 3)
            self.tree_update(R(reg), 5, defer=defer)
            return
        join = self.new("LT")
        e("s_bitcmp1_b32 {ps}, 3")
        e("s_cbranch_scc1 " + L(tag + "_psb"))
        e("s_add_u32 {sym}, {ps}, 8")
        self.tree_walk(R(reg), 3)
        self.tree_update(R(reg), 6)
        self.lab(join)
        with Gen._Into(self, self.cold2):                 # (not the cold list: the callers may be emitting into it; not the
            self.lab(tag + "_psb")                        #  stubs list: the walk's own normalisation stubs go there)
            e("s_mov_b32 {sym}, {ps}")                    # 8 + (ps - 8)
            self.tree_walk(R(reg_b), 3)
            self.tree_update(R(reg_b), 6)
            e("s_branch " + L(join))

    def len_decode(self, which, done, defer_low=False):
        """which: 0 = len_decoder, 1 = rep_len_decoder.  Result (the match length) in mlen; jumps to `done`.
        defer_low: the low tree's update stays queued and `done` is NOT placed here (the caller places it, out of line)."""
        p = "m_len" if which == 0 else "m_rlen"
        w = "l%d" % which
        choice, choice2 = str(48 + 2 * which), str(49 + 2 * which)
        CH = R("m_rep") if self.pb4 else R("m_ismatch")   # (pb4: is_match fills its registers; m_rep's lanes 48..51 are free)
        self.decide(CH, choice, w + "_nlow")
        self.len_tree3(p + "_low", p + "_low_b", w + "lo", defer=defer_low)
        self.e("s_and_b32 {t0}, {sym}, 7")                # length = 2 + path = 9 - inverted path
        self.e("s_sub_u32 {mlen}, 9, {t0}")
        if not defer_low:
            self.lab(done)                                # the common (short) lengths fall through
        with self.in_cold():
            self.lab(w + "_nlow")
            self.taken(CH)
            self.decide(CH, choice2, w + "_high")
            self.len_tree3(p + "_mid", p + "_mid_b", w + "mi")
            self.e("s_and_b32 {t0}, {sym}, 7")            # length = 10 + path
            self.e("s_sub_u32 {mlen}, 17, {t0}")
            self.e("s_branch " + self.L(done))
            # high: tree of 8, nodes 1..63 in h0, 64..127 in h1, 128..191 in h2, 192..255 in h3
            self.lab(w + "_high")
            self.taken(CH)
            self.tree_walk(R(p + "_h0"), 6, first_lane="1")
            self.tree_update(R(p + "_h0"), 6)
            self.bit(R(p + "_h1"), R("sym"), cmp_lane=V["VLANE64"])
            self.e("s_bitcmp1_b32 {sym}, 6")
            self.e("s_cbranch_scc1 " + self.L(w + "_h3"))
            self.bit(R(p + "_h2"), R("sym"), cmp_lane=V["VLANE128"], defer=False)
            self.lab(w + "_hfin")
            self.e("s_sub_u32 {mlen}, 0x211, {sym}")      # length = 18 + path = 18 + 255 - (sym - 256)
            self.e("s_cmpk_ge_u32 {mlen}, 64")        # copies of 64 bytes and more are done outside the loop
            self.e("s_cselect_b32 {gdist}, 0, {gdist}")
            self.e("s_branch " + self.L(done))
            self.lab(w + "_h3")
            self.bit(R(p + "_h3"), R("sym"), cmp_lane=V["VLANE192"], defer=False)
            self.e("s_branch " + self.L(w + "_hfin"))

    def reverse_tree_based(self, T, base_reg, nbits, out):
        """parse_reverse_bit_tree (rangecoder.rs:136-151), node n at lane base + n"""
        self.e("s_mov_b32 {sym}, 1")
        for _ in range(nbits):
            self.e("s_add_u32 {ln}, {b}, {sym}", b=base_reg)
            self.bit(T, R("ln"))
        self.unreverse(nbits, out)

    def unreverse(self, nbits, out):
        self.e("s_not_b32 {o}, {sym}", o=out)            # low nbits = true path, first bit decoded on top
        self.e("s_lshl_b32 {o}, {o}, %d" % (32 - nbits), o=out)
        self.e("s_brev_b32 {o}, {o}", o=out)

    # ---- the loop ------------------------------------------------------------------------------------------
    def prev_fetch(self, ret):
        e = self.e
        e("s_mov_b32 {pend_n}, 0")
        e("s_mov_b32 {prev}, 0")
        e("s_cmp_eq_u32 {len}, 0")
        e("s_cbranch_scc1 " + self.L(ret))
        e("s_add_u32 {t0}, {len}, -1")
        e("v_mov_b32 {VT0}, {t0}")
        e("buffer_load_ubyte {VT0}, {VT0}, {out_rsrc}, 0 offen")
        e("s_waitcnt vmcnt(0)")
        e("v_readfirstlane_b32 {prev}, {VT0}")
        e("s_branch " + self.L(ret))

    def row_swap_stub_hbm(self, name, ret):
        """hbm variant: park the walked row in its slot, bring the new row's slot up to date (tag check; on a miss: the slot's old
        row back to the slab, the new one in), unpack it"""
        e, L = self.e, self.L
        k = self.new("HS")
        with self.in_cold():
            self.lab(name)
            e("s_and_b32 {t3}, {row}, 7")                       # the new row's slot
            e("s_and_b32 {t0}, {cur_row}, 7")
            if SWAP2:
                e("s_set_gpr_idx_on {t0}, gpr_idx(DST)")
                e("v_lshl_or_b32 " + LIT0 + ", {u1}, 16, {u0}")
                e("v_lshl_or_b32 " + LIT1 + ", {u3}, 16, {u2}")
                e("s_set_gpr_idx_off")
            else:
                e("v_lshl_or_b32 {VT0}, {u1}, 16, {u0}")
                e("v_lshl_or_b32 {VT1}, {u3}, 16, {u2}")
                e("s_set_gpr_idx_on {t0}, gpr_idx(DST)")
                e("v_mov_b32 " + LIT0 + ", {VT0}")
                e("v_mov_b32 " + LIT1 + ", {VT1}")
                e("s_set_gpr_idx_off")
            e("v_readlane_b32 {t1}, {vtag}, {t3}")              # whose row the slot holds
            e("s_cmp_eq_u32 {t1}, {row}")
            e("s_cbranch_scc1 " + L(k + "hit"))
            e("s_cmp_eq_u32 {t1}, -1")
            e("s_cbranch_scc1 " + L(k + "load"))
            e("s_set_gpr_idx_on {t3}, gpr_idx(SRC0)")           # the evicted row goes back (a store nobody waits for)
            e("v_mov_b32 {VT0}, " + LIT0)
            e("v_mov_b32 {VT1}, " + LIT1)
            e("s_set_gpr_idx_off")
            e("s_mul_i32 {t2}, {t1}, 0x600")
            e("v_lshl_add_u32 {VA}, {v_lane}, 3, {t2}")
            e("buffer_store_dwordx2 v[%d:%d], {VA}, {lit_rsrc}, 0 offen" % (int(V["VT0"][1:]), int(V["VT1"][1:])))
            self.lab(k + "load")
            e("s_mul_i32 {t2}, {row}, 0x600")
            e("v_lshl_add_u32 {VA}, {v_lane}, 3, {t2}")
            e("buffer_load_dwordx2 v[%d:%d], {VA}, {lit_rsrc}, 0 offen" % (int(V["VT0"][1:]), int(V["VT1"][1:])))
            e("s_mov_b32 m0, {t3}")
            e("s_waitcnt vmcnt(0)")
            e("v_writelane_b32 {vtag}, {row}, m0")
            e("s_set_gpr_idx_on {t3}, gpr_idx(DST)")
            e("v_mov_b32 " + LIT0 + ", {VT0}")
            e("v_mov_b32 " + LIT1 + ", {VT1}")
            e("s_set_gpr_idx_off")
            e("s_branch " + L(k + "unpack"))
            self.lab(k + "hit")
            if SWAP2:
                e("s_set_gpr_idx_on {t3}, gpr_idx(SRC1)")
                e("v_and_b32 {u0}, 0xffff, " + LIT0)
                e("v_lshrrev_b32 {u1}, 16, " + LIT0)
                e("v_and_b32 {u2}, 0xffff, " + LIT1)
                e("v_lshrrev_b32 {u3}, 16, " + LIT1)
                e("s_set_gpr_idx_off")
                e("s_mov_b32 {cur_row}, {row}")
                e("s_branch " + self.L(ret))
            else:
                e("s_set_gpr_idx_on {t3}, gpr_idx(SRC0)")
                e("v_mov_b32 {VT0}, " + LIT0)
                e("v_mov_b32 {VT1}, " + LIT1)
                e("s_set_gpr_idx_off")
            self.lab(k + "unpack")
            e("v_and_b32 {u0}, 0xffff, {VT0}")
            e("v_lshrrev_b32 {u1}, 16, {VT0}")
            e("v_and_b32 {u2}, 0xffff, {VT1}")
            e("v_lshrrev_b32 {u3}, 16, {VT1}")
            e("s_mov_b32 {cur_row}, {row}")
            e("s_branch " + self.L(ret))

    def row_swap_stub(self, name, ret):
        """out of line: park the cached literal row, unpack the new one (cur_row -> row)"""
        if self.hbm:
            return self.row_swap_stub_hbm(name, ret)
        e = self.e
        if SWAP2 and LITSPLIT:
            with self.in_cold():
                self.lab(name)
                e("s_set_gpr_idx_on {cur_row}, gpr_idx(DST)")       # the walked row, packed, straight into its registers
                e("v_lshl_or_b32 " + LIT0 + ", {u1}, 16, {u0}")
                e("v_lshl_or_b32 " + LIT1 + ", {u3}, 16, {u2}")
                e("s_set_gpr_idx_on {row}, gpr_idx(SRC1)")          # the new one straight out of its own
                e("v_and_b32 {u0}, 0xffff, " + LIT0)
                e("v_lshrrev_b32 {u1}, 16, " + LIT0)
                e("v_and_b32 {u2}, 0xffff, " + LIT1)
                e("v_lshrrev_b32 {u3}, 16, " + LIT1)
                e("s_set_gpr_idx_off")
                e("s_mov_b32 {cur_row}, {row}")
                e("s_branch " + self.L(ret))
            return
        with self.in_cold():
            self.lab(name)
            e("v_lshl_or_b32 {VT0}, {u1}, 16, {u0}")
            e("v_lshl_or_b32 {VT1}, {u3}, 16, {u2}")
            if LITSPLIT:
                e("s_set_gpr_idx_on {cur_row}, gpr_idx(DST)")
            else:
                e("s_lshl_b32 {t0}, {cur_row}, 1")
                e("s_set_gpr_idx_on {t0}, gpr_idx(DST)")
            e("v_mov_b32 " + LIT0 + ", {VT0}")
            e("v_mov_b32 " + LIT1 + ", {VT1}")
            e("s_set_gpr_idx_off")
            if LITSPLIT:
                e("s_set_gpr_idx_on {row}, gpr_idx(SRC0)")
            else:
                e("s_lshl_b32 {t0}, {row}, 1")
                e("s_set_gpr_idx_on {t0}, gpr_idx(SRC0)")
            e("v_mov_b32 {VT0}, " + LIT0)
            e("v_mov_b32 {VT1}, " + LIT1)
            e("s_set_gpr_idx_off")
            e("v_and_b32 {u0}, 0xffff, {VT0}")
            e("v_lshrrev_b32 {u1}, 16, {VT0}")
            e("v_and_b32 {u2}, 0xffff, {VT1}")
            e("v_lshrrev_b32 {u3}, 16, {VT1}")
            e("s_mov_b32 {cur_row}, {row}")
            e("s_branch " + self.L(ret))

    def symbol_top(self, tag):
        """top of a symbol (lzma.rs:435-459) up to the is_match decision; falls through for a literal"""
        e, lab, L = self.e, self.lab, self.L
        lab("top" + tag)
        # one test for "the size is reached" and "the reader may be at EOF": gtop = target while more than a
        # window of input is left (then off < lim for sure), 0 in the last window (every symbol looks closer)
        e("s_cmp_ge_u32 {len}, {gtop}")
        e("s_cbranch_scc1 " + L("Otop_slow" + tag))
        lab("top2" + tag)
        e("s_and_b32 {ps}, {len}, {pbmask}")
        if self.pb4:
            e("s_lshl4_add_u32 {ln}, {state}, {ps}")
            self.decide_by_reg(["m_ismatch", "m_ismatch_b", "m_ismatch_c"], R("ln"), "match" + tag, "lit" + tag)
        else:
            e("s_lshl2_add_u32 {lnm}, {state}, {ps}")
            self.decide(R("m_ismatch"), R("lnm"), "match")
        with self.in_cold():
            if self.pb4:
                lab("match" + tag + "2")
                e("s_branch " + L("match2"))
            lab("Otop_slow" + tag)
            # the wave's turn with this unit is over (time-sliced launches; qtop = all ones otherwise): nothing of this symbol has been
            # looked at, the loop is re-entered at this very top.  (A literal that landed ON the output limit is undone first, below.)
            e("s_cmp_gt_u32 {len}, {out_lim}")
            e("s_cbranch_scc1 " + L("Otop_lim" + tag))
            e("s_cmp_ge_u32 {len}, {qtop}")
            e("s_cbranch_scc1 " + L("Xquantum"))
            lab("Otop_lim" + tag)
            # a literal decoded at len == out_lim < target: append_literal's error (lzbuffer.rs:206-217); its store fell
            # outside the slice or beyond out_len.  (len > out_lim >= target: a last match overshot the size, below.)
            e("s_cmp_lt_u32 {out_lim}, {target}")
            e("s_cbranch_scc0 " + L("Otop_size" + tag))
            e("s_cmp_gt_u32 {len}, {out_lim}")
            e("s_cbranch_scc1 " + L("Xlimit_undo"))
            lab("Otop_size" + tag)
            e("s_cmp_ge_u32 {len}, {target}")
            e("s_cbranch_scc1 " + L("Xdone_size"))
            e("s_cmp_gt_u32 {len}, {safe_len}")           # from here on every match looks closer
            e("s_cselect_b32 {gdist}, 0, {gdist}")
            if EOFWRAP and FEED:   # FEED: fewer than FEED_MARGIN bytes of the view left (lim = 0: the last window, -off of them left; -1: none)
                e("s_bitcmp1_b32 {lc8}, %d" % FEED_BIT)
                e("s_cbranch_scc0 " + L("Onofeed" + tag))
                e("s_cmp_eq_u32 {lim}, -1")
                e("s_cbranch_scc1 " + L("Xneed_input"))
                e("s_sub_u32 {t0}, {lim}, {off}")                 # bytes of the view left: -off in this window, lim beyond it
                e("s_cmp_lt_u32 {t0}, %d" % FEED_MARGIN)
                e("s_cbranch_scc1 " + L("Xneed_input"))
                lab("Onofeed" + tag)
            e("s_cmp_eq_u32 {off}, {lim}")                    # reader at EOF: the stream may be finished
            e("s_cbranch_scc0 " + L("top2" + tag))
            lab("Ofin_check" + tag)                           # unknown size: finished when the reader is at EOF
            e("s_or_b32 {t0}, {code}, {known}")               # and code == 0 (is_finished_ok, rangecoder.rs:48-50)
            e("s_cbranch_scc0 " + L("Xdone_fin"))
            e("s_branch " + L("top2" + tag))

    def literal_epilogue(self):
        e, L = self.e, self.L
        e("s_xor_b32 {prev}, {sym}, 0x1ff")             # (0x100 | inverted path) -> byte
        e("v_mov_b32 {VT0}, {prev}")
        e("v_or_b32 {VT1}, {len}, {VOOB}")
        if WAITPROF2:
            self.prof_begin()
        e("buffer_store_byte {VT0}, {VT1}, {out_rsrc}, 0 offen" + STORE_MOD)
        if WAITPROF2:
            self.prof_end("m")
        e("s_add_u32 {len}, {len}, 1")
        e("s_branch " + L("topL"))

    def literal_tail(self, cold):
        """levels 6 and 7 of a plain literal and its epilogue.  cold: the copy the matched literal's mismatch paths enter (labels
        plain6 / plain7; no deferred updates); else the hot one (level 6's update rides in level 7's shadow)."""
        e, lab, L = self.e, self.lab, self.L
        pfx = "c" if cold else ""
        if cold:
            lab("plain6")           # nodes 64..127 -> u1 (v_readlane uses the low 6 bits of the lane select)
        self.bit(R("u1"), R("sym"), cmp_lane=V["VLANE64"], defer=not cold)
        if cold:
            lab("plain7")           # nodes 128..191 -> u2, 192..255 -> u3
        e("s_bitcmp1_b32 {sym}, 6")          # (set for a byte with bit 7 clear: the fall-through serves ASCII)
        e("s_cbranch_scc0 " + L(pfx + "plain7_lo"))
        self.bit(R("u3"), R("sym"), cmp_lane=V["VLANE192"], defer=False)
        if not cold:
            lab("lit_done")
        self.literal_epilogue()
        with self.in_cold():
            lab(pfx + "plain7_lo")
            self.bit(R("u2"), R("sym"), cmp_lane=V["VLANE128"], defer=False)
            self.literal_epilogue()                      # (its own copy: one taken branch instead of two)

    def mrow_load_hbm(self):
        """hbm variant: the LDS rows are a cache (slot = row & 7, tags in vtagm); MROW = the row's matched sub-tables"""
        e, L = self.e, self.L
        e("s_and_b32 {t0}, {row}, 7")
        e("v_lshl_add_u32 {VA}, {t0}, 10, {VL16}")
        e("s_nop 1")
        e("v_readlane_b32 {t1}, {vtagm}, {t0}")
        e("s_cmp_lg_u32 {t1}, {row}")
        e("s_cbranch_scc1 " + L("Omrow_miss"))
        e("ds_read_b128 " + MROW + ", {VA}")
        e("s_waitcnt lgkmcnt(0)")
        self.lab("lm_b")
        with self.in_cold():
            self.lab("Omrow_miss")
            e("s_cmp_eq_u32 {t1}, -1")
            e("s_cbranch_scc1 " + L("Omrow_get"))
            e("ds_read_b128 " + MROW + ", {VA}")                 # the evicted row goes back to the slab
            e("s_mul_i32 {t1}, {t1}, 0x600")
            e("s_add_u32 {t1}, {t1}, 0x200")
            e("v_lshl_add_u32 {VT2}, {v_lane}, 4, {t1}")
            e("s_waitcnt lgkmcnt(0)")
            e("buffer_store_dwordx4 " + MROW + ", {VT2}, {lit_rsrc}, 0 offen")
            self.lab("Omrow_get")
            e("s_mul_i32 {t2}, {row}, 0x600")                   # (the row's place in the slab: only a miss needs it)
            e("s_add_u32 {t2}, {t2}, 0x200")
            e("v_lshl_add_u32 {VT2}, {v_lane}, 4, {t2}")
            e("buffer_load_dwordx4 " + MROW + ", {VT2}, {lit_rsrc}, 0 offen")
            e("s_mov_b32 m0, {t0}")
            e("s_waitcnt vmcnt(0)")
            e("v_writelane_b32 {vtagm}, {row}, m0")              # (the LDS slot itself is written by mrow_store, after the literal)
            e("s_branch " + L("lm_b"))

    def mrow_request(self):
        """EARLYLDS: the read of mrow_load, issued as soon as `row` is known"""
        if EARLYLDS and not self.hbm:
            self.e("v_lshl_add_u32 {VA}, {row}, 10, {VL16}")
            self.e("ds_read_b128 " + MROW + ", {VA}")

    def mrow_load(self):
        """MROW = the matched sub-tables of literal row `row` (4 dwords per lane)"""
        e, L = self.e, self.L
        if self.hbm:
            return self.mrow_load_hbm()
        if not EARLYLDS:
            e("v_lshl_add_u32 {VA}, {row}, 10, {VL16}")
            e("ds_read_b128 " + MROW + ", {VA}")
        e("s_waitcnt lgkmcnt(0)")

    def mrow_store(self):
        """the row back to its LDS row (hbm variant: its LDS slot)"""
        self.e("ds_write_b128 {VA}, " + MROW)

    def literal_row(self, tag):
        e, L = self.e, self.L
        if self.lp0:
            e("s_lshr_b32 {row}, {prev}, {lc8}")
        else:
            e("s_and_b32 {t0}, {len}, {lpmask}")
            e("s_lshl_b32 {t0}, {t0}, {lc}")
            e("s_lshr_b32 {t1}, {prev}, {lc8}")
            e("s_add_u32 {row}, {t0}, {t1}")
        e("s_cmp_lg_u32 {row}, {cur_row}")
        e("s_cbranch_scc1 " + L("Orow_swap" + tag))
        self.lab("lit_r" + tag)
        self.row_swap_stub("Orow_swap" + tag, "lit_r" + tag)

    def copy_guard1(self):
        """LzCircularBuffer::append_lz (lzbuffer.rs:255-281), short and unclipped: one guard, the previous match's store
        completed inline (72 % of the matches of text follow a match), the load of this one issued"""
        e, lab, L = self.e, self.lab, self.L
        e("s_cmp_ge_u32 {rep0}, {gdist}")                     # gdist <= min(len, dict_size), 0 for long / late matches
        e("s_cbranch_scc1 " + L("Ocopy_dist"))                # (the end marker, rep0 = 0xFFFFFFFF, goes there too)
        lab("cp_lim_ok")
        e("s_cmp_eq_u32 {pend_n}, 0")
        e("s_cbranch_scc1 " + L("cp_a"))
        e("s_cmpk_ge_u32 {pend_n}, 0x%x" % PEND_UNKNOWN)
        e("s_cbranch_scc1 " + L("Opend_clear"))
        # (prev / mb are not taken from it: a match follows, and whatever symbol comes after that gets them
        #  from that match's bytes)
        self.finish_pending(prof="c", extract=False)
        lab("cp_a")
        e("s_sub_u32 {t2}, {len}, {rep0}")                    # src + 1 = pos - rep0
        e("s_cmp_lt_u32 {rep0}, {mlen}")                      # dist <= n: the match overlaps itself
        e("s_cbranch_scc1 " + L("Operiodic"))
        e("v_add_u32 {VT0}, {t2}, {VLANEM1}")
        lab("cp_b")
        e("v_cmp_ge_u32 vcc, {mlen}, {v_lane}")               # lanes 0..n: n bytes + the byte after the source
        e("s_mov_b32 {pend_pos}, {len}")
        e("s_mov_b32 {pend_n}, {mlen}")
        e("s_add_u32 {len}, {len}, {mlen}")
        e("v_cndmask_b32 {VT0}, -1, {VT0}, vcc")
        if WAITPROF2:
            self.prof_begin()
        e("buffer_load_ubyte {pend_val}, {VT0}, {out_rsrc}, 0 offen" + LOAD_MOD)
        if WAITPROF2:
            self.prof_end("c")                               # falls through into the top after a match

    def posslot_writeback(self):
        """the walked pos_slot tree (VPS, its update complete) back to the register of its len_state (t5 = len_state + 2)"""
        self.flush()
        with self.at(sec="pos_slot", role="book"):
            self.e("s_set_gpr_idx_on {t5}, gpr_idx(DST)")
            self.e("v_mov_b32 " + PS0M2 + ", {VPS}")
            self.e("s_set_gpr_idx_off")

    def posslot_head(self):
        """the walked pos_slot tree into VPS, the walk's first two levels"""
        e = self.e
        e("s_min_u32 {t5}, {mlen}, 5")                      # len_state + 2  (gfx9 has no v_movrel*: s_set_gpr_idx)
        e("s_set_gpr_idx_on {t5}, gpr_idx(SRC0)")
        e("v_mov_b32 {VPS}, " + PS0M2)
        e("s_set_gpr_idx_off")
        self.bit_nu(V["VPS"], "1", first=True)
        self.bit_nu(V["VPS"], R("sym"))

    def distance_tables(self, split_head=False):
        """decode_distance (lzma.rs:563-592): pos_slot tree of len_state, then by table (tables_prologue) to the code
        for this slot.  rep0 = tbl_a - ((d' << 4) + a') for slots >= 14 (d', a': the inverted direct / align bits)."""
        e, lab, L = self.e, self.lab, self.L
        self.sec = "pos_slot"
        if not split_head:
            self.posslot_head()
        else:                                               # (the caller emitted the head, twice, and the label behind it)
            pass
        for i in range(2, 6):
            self.bit_nu(V["VPS"], R("sym"))
        self.tree_update(V["VPS"], 6, defer=True)            # (queued: emitted in the shadows of the align walk; the tree goes
        self.sec = "dist dispatch"
        if DISP2:
            e("s_flbit_i32_b32 {t6}, {range}")               # leading zeros of range: 0..7
            e("v_add_u32 {VT0}, {t6}, {tbl_b}")              # low byte: n + clz
            if VB2_INLINE:
                e("v_lshrrev_b32 {VT2}, 8, {tbl_b}")
                e("v_mad_u32_u24 {VT2}, {t6}, {VCH}, {VT2}")
            else:
                e("v_mad_u32_u24 {VT2}, {t6}, {VCH}, {VB2}")     # B2 + clz * stride
            e("v_bfe_u32 {VT1}, {VT0}, 3, 5")
            e("v_mad_i32_i24 {VT0}, {VT1}, {VNDN}, {VT2}")
            if VDIRECT:
                assert not any(x[2] for x in self.q)             # (nothing queued reads vcc: the chains use it)
                e("s_mov_b64 vcc, 0")
                e("v_mov_b32 {vr}, {range}")
                e("v_mov_b32 {vb}, {code}")
                e("v_mov_b32 {va}, 0")
            if QDIRECT:
                assert not any(x[2] for x in self.q)             # (nothing queued reads vcc: the quotient blocks use it)
            if not VDIRECT or VDIRECT_S:
                e("s_mov_b32 {t4}, 0")
            e("v_readlane_b32 {t2}, {tbl_a}, {sym}")
            e("v_readlane_b32 {t3}, {VT0}, {sym}")
        else:
            e("v_readlane_b32 {t2}, {tbl_a}, {sym}")         #  back to its register once it is complete: posslot_writeback)
        if DISP2:
            pass
        elif DIRECT8:
            # entry = offset + ((n + clz) & 7) * chain size - ((n + clz) >> 3) * normalisation size, for every lane
            e("s_flbit_i32_b32 {t6}, {range}")               # leading zeros of range: 0..7
            e("v_add_u32 {VT0}, {t6}, {tbl_b}")
            e("v_and_b32 {VT1}, 7, {VT0}")
            e("v_bfe_u32 {VT2}, {VT0}, 3, 3")
            e("v_lshrrev_b32 {VT0}, 8, {VT0}")
            if DISPMAD:
                e("v_mad_u32_u24 {VT0}, {VT1}, {VCH}, {VT0}")
                e("v_mad_i32_i24 {VT0}, {VT2}, {VNDN}, {VT0}")
            else:
                e("v_mul_u32_u24 {VT1}, (" + L("dend1") + "-" + L("dend0") + "), {VT1}")
                e("v_mul_u32_u24 {VT2}, (" + L("dn_e") + "-" + L("dn_s") + "), {VT2}")
                e("v_add_u32 {VT0}, {VT0}, {VT1}")
                e("v_sub_u32 {VT0}, {VT0}, {VT2}")
            e("s_mov_b32 {t4}, 0")
            e("v_readlane_b32 {t3}, {VT0}, {sym}")
        else:
            e("v_readlane_b32 {t3}, {tbl_b}, {sym}")
            e("s_mov_b32 {t4}, 0")
        e("s_add_u32 " + JPAIR_LO + ", {jb_lo}, {t3}")
        e("s_addc_u32 " + JPAIR_HI + ", {jb_hi}, 0")
        targets = ["dist_small", "dist_rev"] + (["dchain%d" % v for v in range(8)] + ["dtr_small%d" % v for v in range(8)] +
                                                ["dtr_rev%d" % v for v in range(8)] if DIRECT8 else ["direct_chain"])
        for name in targets:   # the computed jump's targets arrive with what is queued here
            self.lstate[name] = self._st()
        e("s_setpc_b64 " + JPAIR)
        self.sec = "direct bits"
        self.no_align = True
        if DIRECT8:
            for v in range(8):
                lab("dchain%d" % v)
                if v >= 2:                                   # (every chain carries four normalisation blocks, so that they are all
                    self.direct_norm(vec=VDIRECT, slot="s_nop 0" if QDIRECT else None)   # the same size: this one is never reached)
                for r in range(25, -1, -1):                  # r = direct bits left after this one
                    mark = v == 0 and r == 25
                    vec = VDIRECT and r >= VDIRECT_S
                    if QDIRECT:
                        lab("dq%d_%d" % (v, r))              # (a quotient block's way back, or its serial fall-back)
                    if mark:
                        lab("db_s")
                    self.direct_bit(R("t4"), test=False, vec=vec)
                    if mark:
                        lab("db_e")
                    if r % 8 == v:
                        slot = None
                        if QDIRECT:
                            j = min(r, 8, 6)                 # bits up to the next normalisation (or the chain's end), six at most at once
                            if j >= QDIRECT:
                                qn = "dQ%d_%d" % (v, r)
                                slot = "s_branch " + L(qn)
                            else:
                                slot = "s_nop 0"
                        self.direct_norm(mark="dn" if (v == 0 and r == 24) else None, vec=vec, slot=slot)
                        if QDIRECT and slot != "s_nop 0":
                            self.direct_quotient(qn, j, "dq%d_%d" % (v, r - j - 1) if r - j - 1 >= 0 else "direct_done", "dq%d_%d" % (v, r - 1))   # (the chain's end: straight on, not by its trampoline)
                    if VDIRECT and VDIRECT_S and r == VDIRECT_S:
                        if v == 0:
                            lab("dt_s")
                        self.direct_cross()
                        if v == 0:
                            lab("dt_e")
                lab("dend%d" % v)
                e("s_branch " + L("direct_done"))
                lab("dtr_small%d" % v)                       # slots below 14 land here (no direct bits; chain = clz)
                e("s_branch " + L("dist_small"))
                lab("dtr_rev%d" % v)
                e("s_branch " + L("dist_rev"))
        else:
            lab("direct_chain")
            for _ in range(26):
                self.direct_bit(R("t4"))
        self.sec = "align"
        self.no_align = False
        lab("direct_done")
        if VDIRECT and not VDIRECT_S:
            self.direct_cross()
        written = False
        for i in range(4):
            if "wb" in SSHADOW and i == 2 and not self.q:    # (the pos_slot tree's update went into the shadows of levels 0 and 1)
                with self.at(sec="pos_slot", role="book"):
                    self.queue_s("s_set_gpr_idx_on {t5}, gpr_idx(DST)", kind="wb")
                    self.queue_s("v_mov_b32 " + PS0M2 + ", {VPS}", kind="wb")
                    self.queue_s("s_set_gpr_idx_off", kind="wb")
                written = True
            self.bit_nu(R("m_align"), "1" if i == 0 else R("sym"), first=(i == 0))
        if SYM_M0:   # (the write-back's s_set_gpr_idx_on overwrites m0: everything that reads the symbol comes first)
            assert not written and not self.lazy
            self.tree_update(R("m_align"), 4)
            e("s_lshl_b32 {t3}, {sym}, 28")
            self.posslot_writeback()
        else:
            if not written:
                self.posslot_writeback()
            if self.lazy:
                with self.at(role="update"):
                    e("s_mov_b32 {asym}, {sym}")
            else:
                self.tree_update(R("m_align"), 4)
            e("s_lshl_b32 {t3}, {sym}, 28")                      # drops the leading 1; the inverted path, first bit on top
        e("s_brev_b32 {t3}, {t3}")                           # a'
        e("s_lshl4_add_u32 {t4}, {t4}, {t3}")
        e("s_sub_u32 {rep0}, {t2}, {t4}")                    # (0xFFFFFFFF = the end marker: caught by copy's distance guard)
        self.sec = "dist slots < 14"
        with self.in_cold():                                 # falls through into `copy`
            lab("dist_small")
            self.posslot_writeback()
            e("s_mov_b32 {rep0}, {t2}")
            e("s_branch " + L("copy"))
            # slots 4..11: pos_decoders[result - slot + node] in m_posdec_a, ndb = 1..4 bits;
            # slots 12, 13: m_posdec_b lanes (slot - 12) * 32 + node, 5 bits
            lab("dist_rev")
            if SYM_M0:
                e("s_xor_b32 {t0}, {sym}, 0x7f")                # pos_slot  (read before the write-back's s_set_gpr_idx_on overwrites m0)
            self.posslot_writeback()
            if not SYM_M0:
                e("s_xor_b32 {t0}, {sym}, 0x7f")                # pos_slot
            e("s_lshr_b32 {t1}, {t0}, 1")
            e("s_add_u32 {t1}, {t1}, -1")                       # num_direct_bits;  t2 = (2 | (slot & 1)) << ndb (table)
            e("s_cmp_lt_u32 {t0}, 12")
            e("s_cbranch_scc0 " + L("dist_rev_b"))
            e("s_sub_u32 {t6}, {t2}, {t0}")
            e("s_mov_b32 {sym}, 1")
            for i in range(1, 5):
                e("s_add_u32 {ln}, {t6}, {sym}")
                self.bit(R("m_posdec_a"), R("ln"))
                if i < 4:
                    e("s_cmp_eq_u32 {t1}, %d" % i)
                    e("s_cbranch_scc1 " + L("dist_rev_fin"))
            lab("dist_rev_fin")
            e("s_not_b32 {t4}, {sym}")
            e("s_sub_u32 {t3}, 32, {t1}")
            e("s_lshl_b32 {t4}, {t4}, {t3}")
            e("s_brev_b32 {t4}, {t4}")
            e("s_add_u32 {rep0}, {t2}, {t4}")
            e("s_branch " + L("copy"))
            lab("dist_rev_b")
            e("s_add_u32 {t6}, {t0}, -12")
            e("s_lshl_b32 {t6}, {t6}, 5")
            self.reverse_tree_based(R("m_posdec_b"), R("t6"), 5, R("t4"))
            e("s_add_u32 {rep0}, {t2}, {t4}")
            e("s_branch " + L("copy"))

    # ---- the loop ------------------------------------------------------------------------------------------
    def build(self):
        e, lab, L = self.e, self.lab, self.L
        # prologue: per-lane constants
        e("v_cmp_eq_u32 vcc, 0, {v_lane}")
        e("v_lshl_add_u32 {VL16}, {v_lane}, 4, {ldsbase}")
        e("v_mov_b32 {VKTOP}, 0x1000000")
        e("v_add_u32 {VLANE64}, 64, {v_lane}")
        e("v_add_u32 {VLANE128}, 0x80, {v_lane}")
        e("v_add_u32 {VLANE192}, 0xc0, {v_lane}")
        e("v_cndmask_b32 {VOOB}, -1, 0, vcc")          # lane 0: 0, others: 0xFFFFFFFF (out of range)
        e("v_mov_b32 {c2017}, 2017")
        e("v_mov_b32 {c2048}, 0x800")
        e("v_add_u32 {VLANEM1}, -1, {v_lane}")
        if K24S:
            assert not PAD_S and not ALIGNLAZY and not (WAITPROF or WAITPROF2)
            e("s_mov_b32 {pad}, 0x1000000")
        if self.lazy:
            e("s_mov_b32 {asym}, 0")
        if DISPMAD and DIRECT8:
            e("v_mov_b32 {VCH}, (" + L("dend1") + "-" + L("dend0") + ")")          # stride of the direct-bit chains
            e("v_mov_b32 {VNDN}, (" + L("dn_s") + "-" + L("dn_e") + ")")           # minus the size of a normalisation block
            if DISP2:                                                                 # ... minus (8 strides + a normalisation block)
                e("v_mov_b32 {VT0}, (" + L("dend0") + "-" + L("dend1") + ")")
                e("v_lshl_add_u32 {VNDN}, {VT0}, 3, {VNDN}")
        if EOFWRAP:      # from the C++ side's reader (aligned windows, off = lane, EOF at lane lim) to the loop's (undone in finish())
            e("s_cmpk_lt_u32 {lim}, 64")
            e("s_cbranch_scc1 " + L("Oentry_last"))
            e("s_add_u32 {off}, {off}, -64")
            e("s_add_u32 {lim}, {lim}, -64")
            e("s_add_u32 {n0}, {lim}, -1")
            e("s_cmpk_lt_u32 {n0}, 63")                      # 1..63 bytes beyond this window: the prefetched window must be the
            e("s_cbranch_scc1 " + L("Oentry_pref"))          # end-aligned one
            lab("entry_ok")
            if NBPRE:
                e("v_readlane_b32 {nb}, {winb}, {off}")
            with self.in_cold():
                lab("Oentry_pref")
                e("s_add_u32 {n0}, {wbase}, {lim}")
                e("v_add_u32 {VR}, {n0}, {v_lane}")
                e("buffer_load_ubyte {winb_next}, {VR}, {in_rsrc}, 0 offen")
                e("s_branch " + L("entry_ok"))
                lab("Oentry_last")                           # EOF inside the current window
                e("s_sub_u32 {n0}, {lim}, {off}")            # bytes left
                e("s_cmp_eq_u32 {n0}, 0")
                e("s_cbranch_scc1 " + L("Oentry_eof"))
                e("s_add_u32 {wbase}, {wbase}, {lim}")       # reload it end-aligned
                e("s_add_u32 {wbase}, {wbase}, -64")
                e("v_add_u32 {VR}, {wbase}, {v_lane}")
                e("buffer_load_ubyte {winb}, {VR}, {in_rsrc}, 0 offen")
                e("s_waitcnt vmcnt(0)")
                e("s_sub_u32 {off}, 0, {n0}")
                e("s_mov_b32 {lim}, 0")
                e("s_branch " + L("entry_ok"))
                lab("Oentry_eof")
                e("s_add_u32 {wbase}, {wbase}, {off}")
                e("s_add_u32 {wbase}, {wbase}, -63")
                e("s_mov_b32 {off}, -1")
                e("s_mov_b32 {lim}, -1")
                e("s_branch " + L("entry_ok"))
        elif OFFBIAS:      # (undone in finish(): outside the loop off is the lane, 0..63)
            e("s_add_u32 {off}, {off}, -64")
            e("s_add_u32 {lim}, {lim}, -64")
        if STATE_TBL:   # lane = state: 0 below 4, state - 3 up to 9, state - 6 from 10 on
            e("v_add_u32 {VSTT}, -3, {v_lane}")
            e("v_max_i32 {VSTT}, 0, {VSTT}")
            e("v_add_u32 {vx}, -6, {v_lane}")
            e("v_cmp_gt_u32 vcc, 10, {v_lane}")
            e("s_nop 0")
            e("s_nop 0")
            e("v_cndmask_b32 {VSTT}, {vx}, {VSTT}, vcc")
        self.set_guards(R("n0"))
        self.tables_prologue()
        if DISP2 and not VB2_INLINE:
            e("v_lshrrev_b32 {VB2}, 8, {tbl_b}")
        if PRIO:
            e("s_getreg_b32 {prioph}, hwreg(HW_REG_HW_ID, 0, 4)")   # this wave's slot on its SIMD
            if PRIO < 0:
                self.set_prio(R("prioph"), R("n0"))
        e("v_ffbh_u32 {VLEVEL}, {v_lane}")               # leading zeros (all ones for lane 0)
        e("v_sub_u32 {VLEVEL}, 31, {VLEVEL}")            # floor(log2(lane)); 32 for lane 0
        e("v_sub_u32 {VSH6}, 6, {VLEVEL}")
        e("v_sub_u32 {VSH6M1}, 5, {VLEVEL}")
        e("v_sub_u32 {VSH5}, 5, {VLEVEL}")
        e("v_sub_u32 {VSH5M1}, 4, {VLEVEL}")
        e("v_sub_u32 {VSH4}, 4, {VLEVEL}")
        e("v_sub_u32 {VSH4M1}, 3, {VLEVEL}")
        # The loop body exists twice up to the literal: "L" after a literal (state < 7, nothing pending,
        # prev at hand: a literal here is a plain one) and "M" after a match (state >= 7: a literal here is
        # a matched one and first completes the pending match).
        e("s_cmpk_ge_u32 {state}, 7")
        e("s_cbranch_scc1 " + L("topM"))
        e("s_cmp_lg_u32 {pend_n}, 0")                    # entered from C++ with prev unknown
        e("s_cbranch_scc1 " + L("Oentry_fix"))

        # ================= after a literal =================
        self.sec = "is_match"
        self.symbol_top("L")
        # ---- plain literal (lzma.rs:526-561)
        self.sec = "literal plain"
        self.literal_row("L")
        if STATE_TBL:
            e("v_readlane_b32 {state}, {VSTT}, {state}")     # state after a literal (lzma.rs:472-478)
        else:
            self.queue_s("s_sub_u32 {state}, {state}, 3", kind="state")    # state after a literal (lzma.rs:472-478), states 0..6
            self.queue_s("s_max_i32 {state}, {state}, 0", kind="state")
        self.tree_walk(R("u0"), 6, first_lane="1")  # nodes 1..63 -> u0
        self.tree_update(R("u0"), 6, defer=True)   # (queued: emitted in the shadows of levels 6 and 7)
        self.literal_tail(False)
        with self.in_cold():    # the same walk entered at level 1..5 by a matched literal after its first mismatch:
            for i in range(1, 6):   # only the levels from pl0 on were walked in u0
                lab("plain%d" % i)
                self.bit_nu(R("u0"), R("sym"))
            self.tree_update(R("u0"), 6, min_level=R("pl0"))
            self.literal_tail(True)   # (its own copy of levels 6 / 7: entered at plain6 / plain7 with nothing queued)

        # Section order: a new match falls through its distance tail into `copy`, and `copy` into the top after a match, so that
        # the only taken branches of a match are the computed jump into the direct bits and the loop's back edge.
        # ================= match (lzma.rs:480-523) =================
        self.sec = "is_match"
        if self.pb4:
            lab("match2")                                     # (the update of is_match was done by the register's own stub)
        else:
            lab("match")
            self.taken(R("m_ismatch"))
        self.sec = "is_rep"
        self.decide(R("m_rep"), R("state"), "rep_match")     # is_rep[state]
        self.queue_s("s_mov_b32 {rep3}, {rep2}", kind="rep")
        self.queue_s("s_mov_b32 {rep2}, {rep1}", kind="rep")
        self.queue_s("s_mov_b32 {rep1}, {rep0}", kind="rep")
        self.sec = "length"
        split = self.split
        self.len_decode(0, "len0_done", defer_low=split)
        for cold in ((False, True) if split else (False,)):
            ctx = self.in_cold() if cold else None
            if ctx:
                ctx.__enter__()
                self.sec = "length"
                lab("len0_done")                            # the rare lengths: nothing queued; their own copy of the walk's head
            self.queue_s("s_cmpk_lt_u32 {state}, 7", kind="state")
            self.queue_s("s_cselect_b32 {state}, 7, 10", kind="state")
            if split:
                self.sec = "pos_slot"
                self.posslot_head()
            if ctx:
                e("s_branch " + L("ps_l2"))
                ctx.__exit__()
        if split:
            assert not self.q, "the low length tree's update did not fit the shadows of the pos_slot walk's head"
            lab("ps_l2")
            if self.lazy:
                with self.at(sec="align"):
                    self.align_pending(True)
        # ---- decode_distance (lzma.rs:563-592)
        self.distance_tables(split_head=split)

        # ================= LZ copy, short and unclipped (lzbuffer.rs:255-281) =================
        self.sec = "copy"
        lab("copy")                                          # mlen = bytes to copy, distance = rep0 + 1
        self.copy_guard1()

        # ================= after a match =================
        self.sec = "is_match"
        self.symbol_top("M")
        self.sec = "literal matched"
        # ---- matched literal: probs[((1 + match_bit) << 8) + sym] (lzma.rs:541-555).  The row's two
        #      matched sub-tables are in LDS, dword k of a lane = nodes 64k..64k+63, low half for
        #      match_bit 0 and high half for match_bit 1.
        e("s_add_u32 {t6}, {pend_n}, -1")                # complete the pending match, unless (rare) there is none
        e("s_cmpk_ge_u32 {t6}, 0x%x" % (PEND_UNKNOWN - 1))  # (pend_n == 0, entering from C++) or prev / mb are unknown
        e("s_cbranch_scc1 " + L("OpendM_special"))
        with self.at(sec="copy"):
            self.finish_pending(have_t6=True, prof="m")
        lab("lit_pM")
        self.literal_row("M")
        self.mrow_request()
        if STATE_TBL:
            e("v_readlane_b32 {state}, {VSTT}, {state}")     # states 7..11 -> 4, 5, 6, 4, 5
        else:
            e("s_cmpk_lt_u32 {state}, 10")                   # states 7..11 -> 4, 5, 6, 4, 5
            e("s_cselect_b32 {t1}, 3, 6")
            e("s_sub_u32 {state}, {state}, {t1}")
        if MLGUARD:
            e("s_cmp_ge_u32 {rep0}, {gdist}")                # gdist <= min(len, dict_size): below it both checks pass
            e("s_cbranch_scc1 " + L("Omlit_dist"))
            lab("lm_dist_ok")
        else:
            e("s_add_u32 {t0}, {rep0}, 1")
            e("s_cbranch_scc1 " + L("Xmatch_dist_dict"))
            e("s_cmp_gt_u32 {t0}, {dict_size}")
            e("s_cbranch_scc1 " + L("Xmatch_dist_dict"))
            e("s_cmp_gt_u32 {t0}, {len}")
            e("s_cbranch_scc1 " + L("Xmatch_dist_out"))
        e("s_cmp_eq_u32 {mb}, -1")
        e("s_cbranch_scc1 " + L("Omb_fetch"))
        lab("lm_a")
        self.mrow_load()
        # levels 0..6 (a mismatch continues in the plain chain): one code chain per value of the match
        # bit, so that a branch is only taken when the match bit differs from the previous level's
        e("s_bitcmp1_b32 {mb}, 7")
        e("s_cbranch_scc1 " + L("lm0_1"))
        for m in (0, 1):
            ctx = None if m == 0 else self.in_cold()
            if ctx:
                ctx.__enter__()
            for i in range(7):
                lab("lm%d_%d" % (i, m))
                first = i == 0
                T = V["M0"] if i < 6 else V["M1"]
                lnreg = "1" if first else R("sym")
                cl = V["VLANE64"] if i == 6 else None
                acc = "s_cselect_b32 {sym}, 3, 2" if first else "s_addc_u32 {sym}, {sym}, {sym}"
                mis = self.new("MIS")
                self.core(T, lnreg, half=m, cmp_lane=cl)
                # still matched if the decoded bit equals the match bit: SCC = (bit == 0)
                e(("s_cbranch_scc0 " if m == 0 else "s_cbranch_scc1 ") + L(mis))
                e(acc)
                self.post_known(T, m == 0, half=m)
                self.norm(kind="lit")
                with Gen._Into(self, self.stubs):         # (not the cold list: the m == 1 chain lives there)
                    lab(mis)
                    e(acc)
                    self.post_known(T, m != 0, half=m)
                    self.mrow_store()
                    if i + 1 < 6:
                        e("s_mov_b32 {pl0}, %d" % (i + 1))
                    self.norm(to="plain%d" % (i + 1), kind="lit")
                if i < 6:
                    e("s_bitcmp1_b32 {mb}, %d" % (7 - (i + 1)))   # the next match bit picks the sub-table
                    e(("s_cbranch_scc1 " if m == 0 else "s_cbranch_scc0 ") + L("lm%d_%d" % (i + 1, 1 - m)))
                elif m == 1:
                    e("s_branch " + L("lm7"))
            if ctx:
                ctx.__exit__()
        lab("lm7")                   # last level: nodes 128..191 in M2, 192..255 in M3; nothing follows a mismatch
        e("s_bitcmp1_b32 {sym}, 6")
        e("s_cbranch_scc1 " + L("lm7_hi"))
        e("s_bitcmp1_b32 {mb}, 0")
        e("s_cbranch_scc1 " + L("lm7_lo_m1"))
        self.core(V["M2"], R("sym"), half=0, cmp_lane=V["VLANE128"])
        self.pre_sym()
        e("s_addc_u32 {sym}, {sym}, {sym}")
        self.post_sym(V["M2"], half=0)
        lab("lm_full")
        self.mrow_store()
        self.norm(to="lit_done", kind="lit")
        with self.in_cold():
            for name, T, half, pre, cl in [("lm7_lo_m1", "M2", 1, None, "VLANE128"), ("lm7_hi", "M3", 0, "lm7_hi_m1", "VLANE192"),
                                           ("lm7_hi_m1", "M3", 1, None, "VLANE192")]:
                lab(name)
                if pre:
                    e("s_bitcmp1_b32 {mb}, 0")
                    e("s_cbranch_scc1 " + L(pre))
                self.core(V[T], R("sym"), half=half, cmp_lane=V[cl])
                self.pre_sym()
                e("s_addc_u32 {sym}, {sym}, {sym}")
                self.post_sym(V[T], half=half)
                e("s_branch " + L("lm_full"))
            lab("OpendM_special")
            e("s_cmp_eq_u32 {pend_n}, 0")
            e("s_cbranch_scc1 " + L("lit_pM"))
            self.prev_fetch("lit_pM")                         # lzb.last_or(0) when the previous byte is not at hand
            lab("Oentry_fix")
            self.prev_fetch("topL")

        # ---- rep matches (lzma.rs:483-509)
        self.sec = "is_rep"
        lab("rep_match")
        self.taken(R("m_rep"))
        e("s_add_u32 {lnm}, {state}, 12")
        self.decide(R("m_rep"), R("lnm"), "rep_123")          # is_rep_g0
        if self.pb4:
            e("s_lshl4_add_u32 {ln}, {state}, {ps}")
            self.decide_by_reg(["m_rep0long", "m_rep0long_b", "m_rep0long_c"], R("ln"), "rep0_long", "rep0_short")
        else:
            e("s_lshl2_add_u32 {lnm}, {state}, {ps}")
            self.decide(R("m_rep0long"), R("lnm"), "rep0_long")   # is_rep_0long
        e("s_cmpk_lt_u32 {state}, 7")                        # short rep
        e("s_cselect_b32 {state}, 9, 11")
        e("s_mov_b32 {mlen}, 1")
        e("s_branch " + L("copy"))
        if self.pb4:
            lab("rep0_long2")
            e("s_branch " + L("rep_len"))
        else:
            lab("rep0_long")
            self.taken(R("m_rep0long"), to="rep_len")
        lab("rep_123")
        self.taken(R("m_rep"))
        e("s_add_u32 {lnm}, {state}, 24")
        self.decide(R("m_rep"), R("lnm"), "rep_23")           # is_rep_g1
        e("s_mov_b32 {t0}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        e("s_mov_b32 {rep0}, {t0}")
        e("s_branch " + L("rep_len"))
        lab("rep_23")
        self.taken(R("m_rep"))
        e("s_add_u32 {lnm}, {state}, 36")
        self.decide(R("m_rep"), R("lnm"), "rep_3")            # is_rep_g2
        e("s_mov_b32 {t0}, {rep2}")
        e("s_mov_b32 {rep2}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        e("s_mov_b32 {rep0}, {t0}")
        e("s_branch " + L("rep_len"))
        lab("rep_3")
        self.taken(R("m_rep"))
        e("s_mov_b32 {t0}, {rep3}")
        e("s_mov_b32 {rep3}, {rep2}")
        e("s_mov_b32 {rep2}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        e("s_mov_b32 {rep0}, {t0}")
        lab("rep_len")
        self.sec = "length"
        self.len_decode(1, "len1_done")
        e("s_cmpk_lt_u32 {state}, 7")
        e("s_cselect_b32 {state}, 8, 11")
        e("s_branch " + L("copy"))

        # ================= out-of-line helpers =================
        self.sec = "copy"
        with self.in_cold():
            lab("Operiodic")                                  # source index = lane % dist (exact: lane < 64)
            e("s_add_u32 {t0}, {rep0}, 1")
            e("s_add_u32 {t2}, {t2}, -1")
            e("v_cvt_f32_u32 {VT1}, {t0}")
            e("v_rcp_f32 {VT1}, {VT1}")
            e("v_cvt_f32_u32 {VT2}, {v_lane}")
            e("v_add_f32 {VT2}, 0.5, {VT2}")
            e("v_mul_f32 {VT2}, {VT2}, {VT1}")
            e("v_cvt_u32_f32 {VT2}, {VT2}")
            e("v_mul_lo_u32 {VT2}, {VT2}, {t0}")
            e("v_sub_u32 {VT2}, {v_lane}, {VT2}")
            e("v_add_u32 {VT0}, {t2}, {VT2}")
            e("s_branch " + L("cp_b"))

            lab("Ocopy_dist")                                 # append_lz's two distance errors, in the reference's order
            e("s_cmp_eq_u32 {rep0}, -1")                  #  rep never is, decoding stops at the first one)
            e("s_cbranch_scc1 " + L("Xmarker"))
            e("s_add_u32 {t0}, {rep0}, 1")
            e("s_cmp_gt_u32 {t0}, {dict_size}")
            e("s_cbranch_scc1 " + L("Xlz_dist_dict"))
            e("s_cmp_gt_u32 {t0}, {len}")
            e("s_cbranch_scc1 " + L("Xlz_dist_out"))
            e("s_cmpk_ge_u32 {mlen}, 64")
            e("s_cbranch_scc1 " + L("Xlz_slow"))
            e("s_cmp_gt_u32 {len}, {safe_len}")           # within 273 bytes of the output limit
            e("s_cbranch_scc1 " + L("Ocopy_limit"))
            e("s_min_u32 {gdist}, {len}, {dict_size}")    # the bound was stale
            e("s_branch " + L("cp_lim_ok"))
            lab("Ocopy_limit")                                # (the output resource starts at dict_base: pos = len)
            e("s_add_u32 {t2}, {len}, {mlen}")
            e("s_cbranch_scc1 " + L("Xlz_slow"))
            e("s_cmp_gt_u32 {t2}, {out_lim}")
            e("s_cbranch_scc1 " + L("Xlz_slow"))
            e("s_branch " + L("cp_lim_ok"))
            lab("Opend_copy")
            e("s_cmpk_ge_u32 {pend_n}, 0x%x" % PEND_UNKNOWN)
            e("s_cbranch_scc1 " + L("Opend_clear"))
            # (prev / mb are not taken from it: a match follows, and whatever symbol comes after that gets them
            #  from that match's bytes)
            self.finish_pending(prof="c", extract=False)
            e("s_branch " + L("cp_a"))
            lab("Opend_clear")
            e("s_mov_b32 {pend_n}, 0")
            e("s_branch " + L("cp_a"))

            self.sec = "refill"
            lab("refill")                                     # subroutine: the window's 64 bytes are used up
            if EOFWRAP:
                e("s_add_u32 {n0}, {lim}, 1")                 # lim is 0 (nothing beyond this window: the reader is at EOF now) or -1
                e("s_cmpk_lt_u32 {n0}, 2")                    # (it was already)
                e("s_cbranch_scc1 " + L("Orefill_end"))
                e("s_waitcnt vmcnt(0)")
                e("v_mov_b32 {winb}, {winb_next}")
                e("s_min_u32 {n1}, {lim}, 64")                # the next window: 64 bytes, or the last ones end-aligned
                e("s_add_u32 {wbase}, {wbase}, {n1}")
                e("s_sub_u32 {off}, 0, {n1}")
                e("s_sub_u32 {lim}, {lim}, {n1}")
                self.set_guards(R("n0"))
                e("s_min_u32 {n1}, {lim}, 64")
                e("s_add_u32 {n0}, {wbase}, {n1}")
            else:
                e("s_waitcnt vmcnt(0)")
                e("v_mov_b32 {winb}, {winb_next}")
                e("s_add_u32 {wbase}, {wbase}, 64")
                e("s_mov_b32 {off}, -64" if OFFBIAS else "s_mov_b32 {off}, 0")
                e("s_sub_u32 {lim}, {lim}, 64")
                self.set_guards(R("n0"))
                e("s_add_u32 {n0}, {wbase}, 64")
            e("v_add_u32 {VR}, {n0}, {v_lane}")
            e("buffer_load_ubyte {winb_next}, {VR}, {in_rsrc}, 0 offen")
            if PRIO > 0 and PRIO_TIME:
                if PRIO_LAST:
                    # bit 8 of prioph: "the launch's last block has started" has been seen.  Until then the flag is polled at
                    # every 16th refill only (a coherent scalar load of one word that 4096 waves share is slow when hammered).
                    e("s_bitcmp1_b32 {prioph}, 8")
                    e("s_cbranch_scc1 " + L("rot"))
                    e("s_and_b32 {n1}, {wbase}, 0x3c0")
                    e("s_cbranch_scc1 " + L("norot"))
                    e("s_load_dword {n1}, {flagptr}, 0x0 glc")     # (n1: free again after the test above; s96 holds asym)
                    e("s_waitcnt lgkmcnt(0)")
                    e("s_cmp_eq_u32 {n1}, 0")
                    e("s_cbranch_scc1 " + L("norot"))
                    e("s_bitset1_b32 {prioph}, 8")
                    lab("rot")
                e("s_memtime s[94:95]")
                e("s_waitcnt lgkmcnt(0)")
                e("s_lshr_b64 s[94:95], s[94:95], %d" % PRIO_TIME)
                e("s_add_u32 {n1}, s94, {prioph}")
                self.set_prio(R("n1"), R("n0"))
                if PRIO_LAST:
                    lab("norot")
            elif PRIO > 0:
                e("s_lshr_b32 {n1}, {len}, %d" % PRIO)       # (n0 / n1: the only temporaries free wherever a refill happens)
                e("s_add_u32 {n1}, {n1}, {prioph}")
                self.set_prio(R("n1"), R("n0"))
            if NBPRE:
                e("v_readlane_b32 {nb}, {winb}, {off}")
            e("s_setpc_b64 " + RET)
            if EOFWRAP:
                lab("Orefill_end")
                e("s_cmp_eq_u32 {lim}, -1")                   # the byte just taken was already beyond the end: normalize()'s
                e("s_cbranch_scc1 " + L("Xeof"))              # read error (rangecoder.rs:64)
                e("s_mov_b32 {lim}, -1")                      # the last byte was taken: from now on the reader is at EOF
                e("s_mov_b32 {off}, -1")
                e("s_add_u32 {wbase}, {wbase}, 1")            # (position = wbase + 64 + off stays)
                self.set_guards(R("n0"))
                e("s_setpc_b64 " + RET)

            self.sec = "literal matched"
            if MLGUARD:
                lab("Omlit_dist")                             # the exact checks, in the reference's order; then a fresh bound if one is due
                e("s_add_u32 {t0}, {rep0}, 1")
                e("s_cbranch_scc1 " + L("Xmatch_dist_dict"))
                e("s_cmp_gt_u32 {t0}, {dict_size}")
                e("s_cbranch_scc1 " + L("Xmatch_dist_dict"))
                e("s_cmp_gt_u32 {t0}, {len}")
                e("s_cbranch_scc1 " + L("Xmatch_dist_out"))
                e("s_cmpk_ge_u32 {mlen}, 64")                 # (gdist = 0 on purpose: the copy of a long match leaves the loop)
                e("s_cbranch_scc1 " + L("lm_dist_ok"))
                e("s_cmp_gt_u32 {len}, {safe_len}")           # (... and so does everything near the output limit)
                e("s_cbranch_scc1 " + L("lm_dist_ok"))
                e("s_min_u32 {gdist}, {len}, {dict_size}")
                e("s_branch " + L("lm_dist_ok"))
            lab("Omb_fetch")                                  # lzb.last_n(rep0 + 1)
            if MLGUARD:
                e("s_add_u32 {t0}, {rep0}, 1")
            e("s_sub_u32 {t1}, {len}, {t0}")
            e("v_mov_b32 {VT0}, {t1}")
            e("buffer_load_ubyte {VT0}, {VT0}, {out_rsrc}, 0 offen")
            e("s_waitcnt vmcnt(0)")
            e("v_readfirstlane_b32 {mb}, {VT0}")
            e("s_branch " + L("lm_a"))

            self.sec = "exit"
            lab("Xlimit_undo")
            e("s_add_u32 {len}, {len}, -1")
            self.exit_with("LIMIT")
            for name, code in [("Xdone_size", "DONE_SIZE"), ("Xdone_fin", "DONE_FIN"), ("Xeof", "INPUT_EOF"),
                               ("Xmarker", "MARKER"), ("Xlimit", "LIMIT"), ("Xlz_slow", "LZ_SLOW"),
                               ("Xmatch_dist_dict", "MATCH_DIST_DICT"), ("Xmatch_dist_out", "MATCH_DIST_OUT"),
                               ("Xlz_dist_dict", "LZ_DIST_DICT"), ("Xlz_dist_out", "LZ_DIST_OUT"), ("Xquantum", "QUANTUM"),
                               ("Xneed_input", "NEED_INPUT")]:
                lab(name)
                self.exit_with(code)

    def finish(self):
        # common exit: complete the pending match so that the C++ side sees memory and prev/mb up to date
        e, lab, L = self.e, self.lab, self.L
        lab("finish")
        if self.lazy:
            self.align_pending(False)
        e("s_cmp_eq_u32 {pend_n}, 0")
        e("s_cbranch_scc1 " + L("finish2"))
        e("s_cmpk_ge_u32 {pend_n}, 0x%x" % PEND_UNKNOWN)
        e("s_cbranch_scc1 " + L("finish2"))
        self.finish_pending()
        lab("finish2")
        if EOFWRAP:
            # back to lanes: position = wbase + off, bytes left = lim - off (EOF state: 63 - 63).  Bit 8 of the exit code: the window
            # pair is not the aligned one the C++ side reads from (last window end-aligned / reader at EOF): it re-seeks before it reads
            e("s_add_u32 {n0}, {lim}, 1")
            e("s_cmpk_lt_u32 {n0}, 65")                     # lim in -1 .. 63
            e("s_cselect_b32 {n0}, 0x100, 0")
            e("s_or_b32 {exitcode}, {exitcode}, {n0}")
        if OFFBIAS:
            e("s_add_u32 {off}, {off}, 64")
            e("s_add_u32 {lim}, {lim}, 64")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")


# MIXV (tuning build, round 6: VERDICT r5 item 2d "per-wave mixing"): a fifth instance of the loop, LP0V = LP0 with the tree walks' range < 2^24
# test on the vector ALU (v_cmp + s_cbranch_vccnz: one scalar instruction less per tree decision, one vector instruction more).  The kernel built
# with -DMILZMA_MIXV runs it on the waves in slot 0 of their SIMD (one wave in four), the other three keep the scalar test: the four waves of a
# SIMD share its vector pipe and its turn on the CU's scalar pipe, so a mix of forms shifts load between the pipes without every wave paying.
MIXV = os.environ.get("MILZMA_GEN_MIXV", "0") == "1"


def main():
    global NORM_S, VB2_INLINE
    texts, clobbers, fixeds = {}, {}, {}
    variants = ([("LP0", True, False), ("GEN", False, False), ("PB4", False, True), ("HBM", False, True), ("HB0", True, False)] +
                ([("LP0V", True, False)] if MIXV else []))
    for name, lp0, pb4 in variants:
        saved = (NORM_S, VB2_INLINE)
        if name == "LP0V":
            NORM_S = NORM_S - {"tree"}
            VB2_INLINE = DISP2 and not NORM_S >= {"tree", "single", "lit", "direct"}
        g = Gen(lp0, pb4, hbm=(name in ("HBM", "HB0")))
        g.build()
        NORM_S, VB2_INLINE = saved
        lines = g.main + g.cold + g.cold2 + g.stubs
        g.cur = lines
        g.finish()
        texts[name] = lines
        clobbers[name] = list(CLOBBER_V)
        fixeds[name] = (['"+{v%d}"(d.lit[%d])' % (VBASE + i, i) for i in range(LIT_REGS)] +
                        ['"+{v%d}"(d.posslot[%d])' % (int(PS0[1:]) + i, i) for i in range(4)])
    out = []
    out.append("// GENERATED by tools/gen_fast_loop.py -- do not edit; edit the generator and re-run it.")
    out.append("// The symbol loop of decode_fast_asm_kernel as one inline-asm statement (see the generator's docstring):")
    out.append("// MILZMA_FAST_LOOP_TEXT_LP0 for lp == 0, _GEN for any lp, _PB4 for pb 3 / 4 (any lp) -- lc + lp <= 3 --, _HBM for lc + lp >= 4 (any pb:")
    out.append("// the literal rows in a slab in memory), _HB0 for lc >= 4 with lp == 0 and pb <= 2 (the same slab behind the LP0 variant's cheaper")
    out.append("// bookkeeping); one operand list for all of them.")
    out.append("// clang-format off")
    for k, v in EXIT.items():
        out.append("#define MILZMA_LOOP_EXIT_%s %du" % (k, v))
    out.append("#define MILZMA_LOOP_PEND_UNKNOWN 0x%xu" % PEND_UNKNOWN)
    out.append("#define MILZMA_LOOP_EXIT_RESEEK 0x100u   /* or-ed into the exit code: reload the input windows before reading on */")
    out.append("#define MILZMA_LOOP_FEED_BIT 0x%xu   /* or-ed into the lc8 operand: leave (NEED_INPUT) at the first symbol top with fewer than %d bytes of the input view left */" % (1 << FEED_BIT, FEED_MARGIN))
    for name, lines in texts.items():
        out.append("#define MILZMA_FAST_LOOP_TEXT_%s \\" % name)
        if ALIGN:
            out.append('  ".p2align %d\\n\\t" \\' % ALIGN)
            for _ in range(ALIGN_PAD):
                out.append('  "s_nop 0\\n\\t" \\')
        for l in lines:
            out.append('  "%s\\n\\t" \\' % l.strip())
        out.append('  ""')
    assert clobbers["LP0"] == clobbers["GEN"] and fixeds["LP0"] == fixeds["GEN"]
    assert clobbers["LP0"] == clobbers["PB4"] and fixeds["LP0"] == fixeds["PB4"]
    assert clobbers["LP0"] == clobbers["HBM"] and fixeds["LP0"] == fixeds["HBM"]
    assert clobbers["LP0"] == clobbers["HB0"] and fixeds["LP0"] == fixeds["HB0"]

    def common(vnames):
        return ([('"+{%s}"(d.%s)' % (FIXED_OPERANDS[n], n)) if n in FIXED_OPERANDS else ('[%s] "+s"(d.%s)' % (n, n)) for n in OPS_INOUT_S] +
                [('[%s] "+{v%d}"(d.%s)' % (n, PINV + i, n)) if PINV else ('[%s] "+v"(d.%s)' % (n, n)) for i, n in enumerate(vnames)])
    ins = ['[%s] "s"(d.%s)' % (n, n) for n in OPS_IN_S] + ['[%s] "v"(d.%s)' % (n, n) for n in OPS_IN_V]
    out.append("#define MILZMA_FAST_LOOP_OUTPUTS \\")
    out.append("  " + ", \\\n  ".join(common(OPS_INOUT_V) + fixeds["LP0"]))
    out.append("#define MILZMA_FAST_LOOP_CLOBBERS \\")
    out.append("  " + ", ".join('"%s"' % c for c in CLOBBER_S + clobbers["LP0"]) + ', "vcc", "scc", "memory"')
    out.append("#define MILZMA_FAST_LOOP_INPUTS \\")
    out.append("  " + ", \\\n  ".join(ins))
    out.append("#define MILZMA_FAST_LOOP_UNIFORM(d, rf) \\")
    out.append("  " + " \\\n  ".join("d.%s = rf(d.%s);" % (n, n) for n in OPS_INOUT_S))
    out.append("// clang-format on")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lzma_rs_amd", "csrc", "fast_loop_asm.inc")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    for name, lines in texts.items():
        print("%s: %d instructions" % (name, sum(1 for l in lines if not l.endswith(":"))))
    print("wrote " + os.path.normpath(path))


if __name__ == "__main__":
    main()
