#!/usr/bin/env python3
"""Generates lzma_rs_amd/csrc/fast_loop_asm.inc: the LZMA symbol loop of the fast kernel as ONE gfx950
inline-asm statement (text + operand lists), so that nothing in the hot loop is left to hipcc's
register allocator / block placement (rocprof on the C++ loop: 45 % of issued instructions were
phi copies and other compiler glue).

What the loop does is DecoderState::process_mode(Finish) (src/decode/lzma.rs:435-524) with
decode_literal (526-561), decode_distance (563-592), LenDecoder::decode (rangecoder.rs:256-269),
RangeDecoder::decode_bit / get_bit / normalize (rangecoder.rs:59-120) and the short-match part of
LzCircularBuffer::append_lz (lzbuffer.rs:255-281).  Everything rare leaves the loop with an exit code
and is finished by the C++ around it (decode_fast_asm.hip.h): errors, long or clipped matches, the
end-of-stream marker, the output limit.

Conventions inside the loop
  * range, code and every piece of LZMA state are SGPRs; the probability model is lane-resident
    (one probability per lane of a VGPR, see decode_fast_asm.hip.h for the layout).
  * A decision:  v_readlane p; bound = (range >> 11) * p; s_sub code - bound sets SCC = (code < bound)
    = "bit is 0"; three s_cselect pick range / code / the update constant; the owning lane's
    probability is updated by v_cmp_eq(lane) + v_cndmask, without touching EXEC.
  * Symbols are accumulated with s_addc, i.e. with INVERTED bits (SCC = bit is 0).  Tree nodes are
    therefore stored at the lane of the inverted path, which is a permutation inside each tree level
    and costs nothing (all probabilities start equal); decoded values are un-inverted once per symbol.
  * Normalisation (1 in ~8 decisions) is an out-of-line stub per site; taken branches only stall the
    wave that takes them, while every extra ALU instruction costs issue bandwidth all 16 waves of
    the CU compete for -- so rare work is always moved behind a branch.
"""
import os
import sys

# ---- physical temporaries (listed as clobbers; never live across the asm statement) ----------------
S = dict(sp="s72", sb="s73", sr1="s74", sc1="s75", sk="s76", ln="s77", sym="s78", n0="s79", n1="s80",
         ps="s81", row="s82", t0="s83", t1="s84", t2="s85", t3="s86", t4="s87", t5="s88", t6="s89",
         q0="s90", q1="s91")
SPAIR = "s[90:91]"  # q0:q1 as a 64-bit temporary
V = dict(M0="v84", M1="v85", M2="v86", M3="v87", VT0="v88", VT1="v89", VT2="v90", VA="v91", VPS="v92",
         vt="v93", VL4="v94", VL16="v95", VOOB="v96")
MROW = "v[84:87]"
LIT0, LIT1 = "v64", "v65"   # literal plain table: 16 dwords v64..v79 (fixed, indexed with s_set_gpr_idx)
PS0 = "v80"                 # pos_slot trees for len_state 0..3: v80..v83

EXIT = dict(DONE_SIZE=0, DONE_FIN=1, INPUT_EOF=2, MARKER=3, LIMIT=4, LZ_SLOW=5, MATCH_DIST_DICT=6,
            MATCH_DIST_OUT=7, LZ_DIST_DICT=8, LZ_DIST_OUT=9)

# ---- operands -------------------------------------------------------------------------------------------
# (name, constraint, C++ expression)
OPS_INOUT_S = ["range", "code", "off", "rem", "wbase", "len", "state", "rep0", "rep1", "rep2", "rep3", "prev",
               "mb", "pend_n", "pend_pos", "cur_row", "mlen", "exitcode"]
OPS_INOUT_V = ["m_ismatch", "m_rep", "m_rep0long", "m_posdec_a", "m_posdec_b", "m_len_lm", "m_len_h0", "m_len_h1",
               "m_len_h2", "m_len_h3", "m_rlen_lm", "m_rlen_h0", "m_rlen_h1", "m_rlen_h2", "m_rlen_h3", "u0",
               "u1", "u2", "u3", "win", "win_next", "pend_val"]
OPS_IN_S = ["out_lim", "target", "known", "dict_size", "dict_base", "lc", "lc8", "lpmask", "pbmask", "in_rsrc",
            "out_rsrc", "lut", "ldsbase"]
OPS_IN_V = ["v_lane"]


def R(name):
    if name in S:
        return S[name]
    if name in V:
        return V[name]
    return "%%[%s]" % name


class Gen:
    def __init__(self):
        self.main, self.cold, self.stubs = [], [], []
        self.cur = self.main
        self.uid = 0

    def e(self, fmt, **kw):
        """emit one instruction; {x} is replaced by the register of operand/temporary x"""
        names = set()
        out = fmt
        import re
        for m in re.findall(r"\{(\w+)\}", fmt):
            names.add(m)
        for n in names:
            out = out.replace("{%s}" % n, kw[n] if n in kw else R(n))
        self.cur.append("  " + out)

    def lab(self, name):
        self.cur.append("L%s%%=:" % name)

    @staticmethod
    def L(name):
        return "L%s%%=" % name

    def new(self, prefix):
        self.uid += 1
        return "%s%d" % (prefix, self.uid)

    class _Into:
        def __init__(self, g, lst):
            self.g, self.lst = g, lst

        def __enter__(self):
            self.saved = self.g.cur
            self.g.cur = self.lst

        def __exit__(self, *a):
            self.g.cur = self.saved

    def in_list(self, lst):
        return Gen._Into(self, lst)

    def in_cold(self):
        return Gen._Into(self, self.cold)

    # ---- range coder ------------------------------------------------------------------------------
    def norm(self, to=None):
        """RangeDecoder::normalize (rangecoder.rs:59-69) as a check + out-of-line stub.
        `to`: label to continue at (default: fall through)."""
        k = self.new("N")
        self.e("s_lshr_b32 {n0}, {range}, 24")          # SCC = range >= 2^24
        self.e("s_cbranch_scc0 " + self.L(k))
        if to is None:
            ret = self.new("R")
            self.lab(ret)
        else:
            ret = to
            self.e("s_branch " + self.L(to))
        with self.in_list(self.stubs):
            self.lab(k)
            self.e("s_cmp_eq_u32 {rem}, 0")
            self.e("s_cbranch_scc1 " + self.L("Xeof"))
            self.e("s_lshl_b32 {range}, {range}, 8")
            self.e("s_lshr_b32 {n0}, {off}, 2")
            self.e("v_readlane_b32 {n1}, {win}, {n0}")
            self.e("s_lshl_b32 {n0}, {off}, 3")          # shift amount uses bits 4:0 = (off & 3) * 8
            self.e("s_lshr_b32 {n1}, {n1}, {n0}")
            self.e("s_and_b32 {n1}, {n1}, 0xff")
            self.e("s_lshl_b32 {code}, {code}, 8")
            self.e("s_or_b32 {code}, {code}, {n1}")
            self.e("s_add_u32 {off}, {off}, 1")
            self.e("s_add_u32 {rem}, {rem}, -1")
            self.e("s_branch " + self.L(ret))

    def bitcore(self, T, ln, half=None):
        """decode_bit (rangecoder.rs:92-120) on the probability in lane `ln` of T, up to the point where
        SCC = (bit == 0).  half: None = T holds one probability per lane; 0 / 1 = low / high 16 bits."""
        self.e("v_readlane_b32 {sp}, {T}, {ln}", T=T, ln=ln)
        self.e("v_cmp_eq_u32 vcc, {ln}, {v_lane}", ln=ln)
        if half == 0:
            self.e("s_and_b32 {sp}, {sp}, 0xffff")
        elif half == 1:
            self.e("s_lshr_b32 {sp}, {sp}, 16")
        self.e("s_lshr_b32 {sb}, {range}, 11")
        self.e("s_mul_i32 {sb}, {sb}, {sp}")
        self.e("s_sub_u32 {sr1}, {range}, {sb}")
        self.e("s_sub_u32 {sc1}, {code}, {sb}")          # SCC = code < bound  <=>  bit == 0
        self.e("s_cselect_b32 {range}, {sb}, {sr1}")
        self.e("s_cselect_b32 {code}, {code}, {sc1}")
        self.e("s_cselect_b32 {sk}, 0x800, 31")          # p += ((bit ? 31 : 2048) - p) >> 5  (arithmetic)
        if half is None:
            self.e("v_sub_u32 {vt}, {sk}, {T}", T=T)
        elif half == 0:
            self.e("v_and_b32 {vt}, 0xffff, {T}", T=T)
            self.e("v_sub_u32 {vt}, {sk}, {vt}")
        else:
            self.e("v_lshrrev_b32 {vt}, 16, {T}", T=T)
            self.e("v_sub_u32 {vt}, {sk}, {vt}")
        self.e("v_ashrrev_i32 {vt}, 5, {vt}")
        if half == 1:
            self.e("v_lshlrev_b32 {vt}, 16, {vt}")
        self.e("v_add_u32 {vt}, {T}, {vt}", T=T)        # low half: the sign extension of a negative delta cancels
        self.e("v_cndmask_b32 {T}, {T}, {vt}, vcc", T=T)  # against the carry out of the (non-negative) low half

    def bit(self, T, ln, half=None):
        """one tree decision: sym = 2 * sym + (bit == 0), then normalise"""
        self.bitcore(T, ln, half)
        self.e("s_addc_u32 {sym}, {sym}, {sym}")
        self.norm()

    def direct_bit(self, acc):
        """RangeDecoder::get_bit (rangecoder.rs:71-82): acc = 2 * acc + (bit == 0)"""
        self.e("s_lshr_b32 {range}, {range}, 1")
        self.e("s_sub_u32 {sc1}, {code}, {range}")
        self.e("s_cselect_b32 {code}, {code}, {sc1}")
        self.e("s_addc_u32 {a}, {a}, {a}", a=acc)
        self.norm()

    # ---- pending short match ---------------------------------------------------------------------------
    def finish_pending(self):
        self.e("s_waitcnt vmcnt(0)")
        self.e("v_cmp_gt_u32 vcc, {pend_n}, {v_lane}")
        self.e("v_add_u32 {VT0}, {pend_pos}, {v_lane}")
        self.e("v_cndmask_b32 {VT0}, -1, {VT0}, vcc")
        self.e("buffer_store_byte {pend_val}, {VT0}, {out_rsrc}, 0 offen")
        self.e("s_add_u32 {t6}, {pend_n}, -1")
        self.e("v_readlane_b32 {prev}, {pend_val}, {t6}")
        self.e("v_readlane_b32 {mb}, {pend_val}, {pend_n}")
        self.e("s_mov_b32 {pend_n}, 0")

    def exit_with(self, code):
        self.e("s_mov_b32 {exitcode}, %d" % EXIT[code])
        self.e("s_branch " + self.L("finish"))

    # ---- LenDecoder::decode ------------------------------------------------------------------------------
    def len_decode(self, which, done):
        """which: 0 = len_decoder, 1 = rep_len_decoder.  Result (length - 2) in mlen; jumps to `done`."""
        p = "m_len" if which == 0 else "m_rlen"
        w = "l%d" % which
        lm = R(p + "_lm")
        choice, choice2 = 48 + 2 * which, 49 + 2 * which
        self.e("s_mov_b32 {ln}, %d" % choice)
        self.bitcore(R("m_ismatch"), R("ln"))
        self.e("s_cbranch_scc0 " + self.L(w + "_nlow"))
        self.norm()
        # low: tree of 3 at lanes ps*8 + node
        self.e("s_lshl_b32 {t0}, {ps}, 3")
        self.lab(w + "_tree3")
        self.e("s_mov_b32 {sym}, 1")
        for _ in range(3):
            self.e("s_add_u32 {ln}, {t0}, {sym}")
            self.bit(lm, R("ln"))
        self.e("s_xor_b32 {mlen}, {sym}, 15")            # (8 | inverted path) -> path
        self.e("s_and_b32 {t0}, {t0}, 32")               # mid (base >= 32) adds 8
        self.e("s_lshr_b32 {t0}, {t0}, 2")
        self.e("s_add_u32 {mlen}, {mlen}, {t0}")
        self.e("s_branch " + self.L(done))
        self.lab(w + "_nlow")
        self.norm()
        self.e("s_mov_b32 {ln}, %d" % choice2)
        self.bitcore(R("m_ismatch"), R("ln"))
        self.e("s_cbranch_scc0 " + self.L(w + "_high"))
        self.norm()
        self.e("s_lshl_b32 {t0}, {ps}, 3")
        self.e("s_add_u32 {t0}, {t0}, 32")
        self.e("s_branch " + self.L(w + "_tree3"))
        # high: tree of 8, nodes 1..63 in h0, 64..127 in h1, 128..191 in h2, 192..255 in h3
        self.lab(w + "_high")
        self.norm()
        self.e("s_mov_b32 {sym}, 1")
        for _ in range(6):
            self.bit(R(p + "_h0"), R("sym"))
        self.e("s_and_b32 {ln}, {sym}, 63")
        self.bit(R(p + "_h1"), R("ln"))
        self.e("s_and_b32 {ln}, {sym}, 63")
        self.e("s_bitcmp1_b32 {sym}, 6")
        self.e("s_cbranch_scc1 " + self.L(w + "_h3"))
        self.bit(R(p + "_h2"), R("ln"))
        self.lab(w + "_hdone")
        self.e("s_xor_b32 {mlen}, {sym}, 0x1ff")
        self.e("s_add_u32 {mlen}, {mlen}, 16")
        self.e("s_branch " + self.L(done))
        self.lab(w + "_h3")
        self.bit(R(p + "_h3"), R("ln"))
        self.e("s_branch " + self.L(w + "_hdone"))

    def reverse_tree_fixed(self, T, base_reg, nbits, out):
        """parse_reverse_bit_tree (rangecoder.rs:136-151) with a compile-time bit count"""
        self.e("s_mov_b32 {sym}, 1")
        for _ in range(nbits):
            self.e("s_add_u32 {ln}, {b}, {sym}", b=base_reg)
            self.bit(T, R("ln"))
        self.e("s_not_b32 {o}, {sym}", o=out)            # low nbits = true path, first bit decoded on top
        self.e("s_lshl_b32 {o}, {o}, %d" % (32 - nbits), o=out)
        self.e("s_brev_b32 {o}, {o}", o=out)

    # ---- the loop ------------------------------------------------------------------------------------------
    def build(self):
        e, lab, L = self.e, self.lab, self.L
        # prologue: per-lane constants
        e("v_lshlrev_b32 {VL4}, 2, {v_lane}")
        e("v_lshl_add_u32 {VL16}, {v_lane}, 4, {ldsbase}")
        e("v_cmp_eq_u32 vcc, 0, {v_lane}")
        e("v_cndmask_b32 {VOOB}, -1, 0, vcc")        # lane 0: 0, others: 0xFFFFFFFF (out of range)

        # ================= top of a symbol (lzma.rs:435-459) =================
        lab("top")
        e("s_cmp_ge_u32 {len}, {target}")
        e("s_cbranch_scc1 " + L("Xdone_size"))
        e("s_or_b32 {t0}, {rem}, {code}")
        e("s_or_b32 {t0}, {t0}, {known}")               # unknown size: finished when rem == 0 and code == 0
        e("s_cbranch_scc0 " + L("Xdone_fin"))
        e("s_cmpk_ge_u32 {off}, 0xc0")
        e("s_cbranch_scc1 " + L("Oslide"))
        lab("top_slid")
        e("s_and_b32 {ps}, {len}, {pbmask}")
        e("s_lshl2_add_u32 {ln}, {state}, {ps}")
        self.bitcore(R("m_ismatch"), R("ln"))
        e("s_cbranch_scc0 " + L("match"))
        self.norm()

        # ================= literal (lzma.rs:526-561) =================
        e("s_cmp_lg_u32 {pend_n}, 0")
        e("s_cbranch_scc1 " + L("Opend_lit"))
        lab("lit_p")
        e("s_cmp_eq_u32 {prev}, -1")
        e("s_cbranch_scc1 " + L("Oprev_fetch"))
        lab("lit_q")
        e("s_and_b32 {t0}, {len}, {lpmask}")
        e("s_lshl_b32 {t0}, {t0}, {lc}")
        e("s_lshr_b32 {t1}, {prev}, {lc8}")
        e("s_add_u32 {row}, {t0}, {t1}")
        e("s_cmp_lg_u32 {row}, {cur_row}")
        e("s_cbranch_scc1 " + L("Orow_swap"))
        lab("lit_r")
        e("s_mov_b32 {sym}, 1")
        e("s_cmpk_ge_u32 {state}, 7")
        e("s_cbranch_scc1 " + L("lit_matched"))
        for i in range(6):          # nodes 1..63 -> u0
            if i > 0:
                lab("plain%d" % i)
            self.bit(R("u0"), R("sym"))
        lab("plain6")               # nodes 64..127 -> u1
        e("s_and_b32 {ln}, {sym}, 63")
        self.bit(R("u1"), R("ln"))
        lab("plain7")               # nodes 128..191 -> u2, 192..255 -> u3
        e("s_and_b32 {ln}, {sym}, 63")
        e("s_bitcmp1_b32 {sym}, 6")
        e("s_cbranch_scc1 " + L("plain7_hi"))
        self.bit(R("u2"), R("ln"))
        lab("lit_done")
        e("s_xor_b32 {prev}, {sym}, 0x1ff")             # (0x100 | inverted path) -> byte
        e("s_add_u32 {t0}, {dict_base}, {len}")
        e("s_cmp_ge_u32 {t0}, {out_lim}")
        e("s_cbranch_scc1 " + L("Xlimit"))
        e("v_mov_b32 {VT0}, {prev}")
        e("v_or_b32 {VT1}, {t0}, {VOOB}")
        e("buffer_store_byte {VT0}, {VT1}, {out_rsrc}, 0 offen")
        e("s_add_u32 {len}, {len}, 1")
        e("s_lshl_b32 {t0}, {state}, 2")                 # state after a literal (lzma.rs:472-478): nibble LUT
        e("s_lshr_b64 " + SPAIR + ", {lut}, {t0}")
        e("s_and_b32 {state}, {q0}, 15")
        e("s_branch " + L("top"))
        with self.in_cold():
            lab("plain7_hi")
            self.bit(R("u3"), R("ln"))
            e("s_branch " + L("lit_done"))

        # ---- matched literal: probs[((1 + match_bit) << 8) + sym] (lzma.rs:541-555).  The row's two
        #      matched sub-tables are in LDS, dword k of a lane = nodes 64k..64k+63, low half for
        #      match_bit 0 and high half for match_bit 1.
        lab("lit_matched")
        e("s_add_u32 {t0}, {rep0}, 1")
        e("s_cbranch_scc1 " + L("Xmatch_dist_dict"))
        e("s_cmp_gt_u32 {t0}, {dict_size}")
        e("s_cbranch_scc1 " + L("Xmatch_dist_dict"))
        e("s_cmp_gt_u32 {t0}, {len}")
        e("s_cbranch_scc1 " + L("Xmatch_dist_out"))
        e("s_cmp_eq_u32 {mb}, -1")
        e("s_cbranch_scc1 " + L("Omb_fetch"))
        lab("lm_a")
        e("v_lshl_add_u32 {VA}, {row}, 10, {VL16}")
        e("ds_read_b128 " + MROW + ", {VA}")
        e("s_waitcnt lgkmcnt(0)")
        for i in range(7):          # levels 0..6: a mismatch continues in the plain chain
            lab("lm%d" % i)
            T, lnname = (V["M0"], "sym") if i < 6 else (V["M1"], "ln")
            if i == 6:
                e("s_and_b32 {ln}, {sym}, 63")
            e("s_bitcmp1_b32 {mb}, %d" % (7 - i))           # the match byte's next bit picks the sub-table
            e("s_cbranch_scc1 " + L("lm%d_m1" % i))
            self.bitcore(T, R(lnname), half=0)               # match bit 0: still matched if the bit is 0 (SCC = 1)
            mis0, mis1 = self.new("MIS"), self.new("MIS")
            e("s_cbranch_scc0 " + L(mis0))
            e("s_addc_u32 {sym}, {sym}, {sym}")
            self.norm()                                      # falls through to the next level
            with self.in_cold():
                lab(mis0)
                e("s_addc_u32 {sym}, {sym}, {sym}")
                e("ds_write_b128 {VA}, " + MROW)
                self.norm(to="plain%d" % (i + 1))
                lab("lm%d_m1" % i)
                self.bitcore(T, R(lnname), half=1)           # match bit 1: still matched if the bit is 1 (SCC = 0)
                e("s_cbranch_scc1 " + L(mis1))
                e("s_addc_u32 {sym}, {sym}, {sym}")
                self.norm(to="lm%d" % (i + 1))
                lab(mis1)
                e("s_addc_u32 {sym}, {sym}, {sym}")
                e("ds_write_b128 {VA}, " + MROW)
                self.norm(to="plain%d" % (i + 1))
        lab("lm7")                   # last level: nodes 128..191 in M2, 192..255 in M3; nothing follows a mismatch
        e("s_and_b32 {ln}, {sym}, 63")
        e("s_bitcmp1_b32 {sym}, 6")
        e("s_cbranch_scc1 " + L("lm7_hi"))
        e("s_bitcmp1_b32 {mb}, 0")
        e("s_cbranch_scc1 " + L("lm7_lo_m1"))
        self.bitcore(V["M2"], R("ln"), half=0)
        e("s_addc_u32 {sym}, {sym}, {sym}")
        lab("lm_full")
        e("ds_write_b128 {VA}, " + MROW)
        self.norm(to="lit_done")
        with self.in_cold():
            lab("lm7_lo_m1")
            self.bitcore(V["M2"], R("ln"), half=1)
            e("s_addc_u32 {sym}, {sym}, {sym}")
            e("s_branch " + L("lm_full"))
            lab("lm7_hi")
            e("s_bitcmp1_b32 {mb}, 0")
            e("s_cbranch_scc1 " + L("lm7_hi_m1"))
            self.bitcore(V["M3"], R("ln"), half=0)
            e("s_addc_u32 {sym}, {sym}, {sym}")
            e("s_branch " + L("lm_full"))
            lab("lm7_hi_m1")
            self.bitcore(V["M3"], R("ln"), half=1)
            e("s_addc_u32 {sym}, {sym}, {sym}")
            e("s_branch " + L("lm_full"))

        # ================= match (lzma.rs:480-523) =================
        lab("match")
        self.norm()
        self.bitcore(R("m_rep"), R("state"))               # is_rep[state]
        e("s_cbranch_scc0 " + L("rep_match"))
        self.norm()
        e("s_mov_b32 {rep3}, {rep2}")
        e("s_mov_b32 {rep2}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        self.len_decode(0, "len0_done")
        lab("len0_done")
        e("s_cmpk_lt_u32 {state}, 7")
        e("s_cselect_b32 {state}, 7, 10")
        # ---- decode_distance (lzma.rs:563-592)
        e("s_min_u32 {t5}, {mlen}, 3")                      # len_state
        e("s_set_gpr_idx_on {t5}, gpr_idx(SRC0)")
        e("v_mov_b32 {VPS}, " + PS0)
        e("s_set_gpr_idx_off")
        e("s_mov_b32 {sym}, 1")
        for _ in range(6):
            self.bit(V["VPS"], R("sym"))
        e("s_set_gpr_idx_on {t5}, gpr_idx(DST)")
        e("v_mov_b32 " + PS0 + ", {VPS}")
        e("s_set_gpr_idx_off")
        e("s_xor_b32 {t0}, {sym}, 0x7f")                    # pos_slot
        e("s_mov_b32 {rep0}, {t0}")
        e("s_cmp_lt_u32 {t0}, 4")
        e("s_cbranch_scc1 " + L("copy"))
        e("s_lshr_b32 {t1}, {t0}, 1")
        e("s_add_u32 {t1}, {t1}, -1")                       # num_direct_bits
        e("s_and_b32 {t2}, {t0}, 1")
        e("s_or_b32 {t2}, {t2}, 2")
        e("s_lshl_b32 {t2}, {t2}, {t1}")                    # result = (2 | (slot & 1)) << ndb
        e("s_cmp_lt_u32 {t0}, 14")
        e("s_cbranch_scc1 " + L("dist_rev"))
        # slots >= 14: ndb - 4 direct bits, then the 4-bit align tree (m_rep0long lanes 48 + node)
        e("s_add_u32 {t3}, {t1}, -4")                       # count
        e("s_mov_b32 {t5}, {t3}")
        e("s_mov_b32 {t4}, 0")
        lab("direct4")
        e("s_cmp_lt_u32 {t3}, 4")
        e("s_cbranch_scc1 " + L("direct_tail"))
        for _ in range(4):
            self.direct_bit(R("t4"))
        e("s_add_u32 {t3}, {t3}, -4")
        e("s_branch " + L("direct4"))
        lab("direct_tail")
        e("s_bitcmp1_b32 {t3}, 1")
        e("s_cbranch_scc0 " + L("direct_t1"))
        for _ in range(2):
            self.direct_bit(R("t4"))
        lab("direct_t1")
        e("s_bitcmp1_b32 {t3}, 0")
        e("s_cbranch_scc0 " + L("direct_done"))
        self.direct_bit(R("t4"))
        lab("direct_done")
        e("s_bfm_b32 {t5}, {t5}, 0")                         # (1 << count) - 1
        e("s_andn2_b32 {t4}, {t5}, {t4}")                    # un-invert
        e("s_lshl_b32 {t4}, {t4}, 4")
        e("s_add_u32 {t2}, {t2}, {t4}")
        e("s_mov_b32 {t6}, 48")
        self.reverse_tree_fixed(R("m_rep0long"), R("t6"), 4, R("t4"))
        e("s_add_u32 {rep0}, {t2}, {t4}")
        e("s_cmp_eq_u32 {rep0}, -1")
        e("s_cbranch_scc1 " + L("Xmarker"))
        e("s_branch " + L("copy"))
        with self.in_cold():
            # slots 4..11: pos_decoders[result - slot + node] in m_posdec_a, ndb = 1..5 bits;
            # slots 12, 13: m_posdec_b lanes (slot - 12) * 32 + node, 5 bits
            lab("dist_rev")
            e("s_cmp_lt_u32 {t0}, 12")
            e("s_cbranch_scc0 " + L("dist_rev_b"))
            e("s_sub_u32 {t6}, {t2}, {t0}")
            e("s_mov_b32 {sym}, 1")
            for i in range(1, 5):                            # slots 4..11 have 1..4 direct bits
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
            self.reverse_tree_fixed(R("m_posdec_b"), R("t6"), 5, R("t4"))
            e("s_add_u32 {rep0}, {t2}, {t4}")
            e("s_branch " + L("copy"))

        # ---- rep matches (lzma.rs:483-509)
        lab("rep_match")
        self.norm()
        e("s_add_u32 {ln}, {state}, 12")
        self.bitcore(R("m_rep"), R("ln"))                   # is_rep_g0
        e("s_cbranch_scc0 " + L("rep_123"))
        self.norm()
        e("s_lshl2_add_u32 {ln}, {state}, {ps}")
        self.bitcore(R("m_rep0long"), R("ln"))              # is_rep_0long
        e("s_cbranch_scc0 " + L("rep0_long"))
        self.norm()
        e("s_cmpk_lt_u32 {state}, 7")                        # short rep
        e("s_cselect_b32 {state}, 9, 11")
        e("s_mov_b32 {mlen}, 1")
        e("s_branch " + L("copy_n"))
        lab("rep0_long")
        self.norm(to="rep_len")
        lab("rep_123")
        self.norm()
        e("s_add_u32 {ln}, {state}, 24")
        self.bitcore(R("m_rep"), R("ln"))                   # is_rep_g1
        e("s_cbranch_scc0 " + L("rep_23"))
        self.norm()
        e("s_mov_b32 {t0}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        e("s_mov_b32 {rep0}, {t0}")
        e("s_branch " + L("rep_len"))
        lab("rep_23")
        self.norm()
        e("s_add_u32 {ln}, {state}, 36")
        self.bitcore(R("m_rep"), R("ln"))                   # is_rep_g2
        e("s_cbranch_scc0 " + L("rep_3"))
        self.norm()
        e("s_mov_b32 {t0}, {rep2}")
        e("s_mov_b32 {rep2}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        e("s_mov_b32 {rep0}, {t0}")
        e("s_branch " + L("rep_len"))
        lab("rep_3")
        self.norm()
        e("s_mov_b32 {t0}, {rep3}")
        e("s_mov_b32 {rep3}, {rep2}")
        e("s_mov_b32 {rep2}, {rep1}")
        e("s_mov_b32 {rep1}, {rep0}")
        e("s_mov_b32 {rep0}, {t0}")
        lab("rep_len")
        self.len_decode(1, "len1_done")
        lab("len1_done")
        e("s_cmpk_lt_u32 {state}, 7")
        e("s_cselect_b32 {state}, 8, 11")

        # ================= LZ copy, short and unclipped (lzbuffer.rs:255-281) =================
        lab("copy")
        e("s_add_u32 {mlen}, {mlen}, 2")
        lab("copy_n")                                        # mlen = bytes to copy, distance = rep0 + 1
        e("s_add_u32 {t0}, {rep0}, 1")
        e("s_cmp_gt_u32 {t0}, {dict_size}")
        e("s_cbranch_scc1 " + L("Xlz_dist_dict"))
        e("s_cmp_gt_u32 {t0}, {len}")
        e("s_cbranch_scc1 " + L("Xlz_dist_out"))
        e("s_cmpk_ge_u32 {mlen}, 64")
        e("s_cbranch_scc1 " + L("Xlz_slow"))
        e("s_add_u32 {t1}, {dict_base}, {len}")               # pos
        e("s_add_u32 {t2}, {t1}, {mlen}")
        e("s_cbranch_scc1 " + L("Xlz_slow"))
        e("s_cmp_gt_u32 {t2}, {out_lim}")
        e("s_cbranch_scc1 " + L("Xlz_slow"))
        e("s_cmp_lg_u32 {pend_n}, 0")
        e("s_cbranch_scc1 " + L("Opend_copy"))
        lab("cp_a")
        e("s_sub_u32 {t2}, {t1}, {t0}")                       # src = pos - dist
        e("s_cmp_le_u32 {t0}, {mlen}")
        e("s_cbranch_scc1 " + L("Operiodic"))
        e("v_add_u32 {VT0}, {t2}, {v_lane}")
        lab("cp_b")
        e("v_cmp_ge_u32 vcc, {mlen}, {v_lane}")               # lanes 0..n: n bytes + the byte after the source
        e("v_cndmask_b32 {VT0}, -1, {VT0}, vcc")
        e("buffer_load_ubyte {pend_val}, {VT0}, {out_rsrc}, 0 offen")
        e("s_mov_b32 {pend_pos}, {t1}")
        e("s_mov_b32 {pend_n}, {mlen}")
        e("s_add_u32 {len}, {len}, {mlen}")
        e("s_branch " + L("top"))

        # ================= out-of-line helpers =================
        with self.in_cold():
            lab("Operiodic")                                  # source index = lane % dist (exact: lane < 64)
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

            lab("Opend_lit")
            self.finish_pending()
            e("s_branch " + L("lit_p"))
            lab("Opend_copy")
            self.finish_pending()
            e("s_branch " + L("cp_a"))

            lab("Oslide")                                     # keep >= 64 input bytes ahead of `off`
            e("s_waitcnt vmcnt(0)")
            e("v_mov_b32 {win}, {win_next}")
            e("s_add_u32 {wbase}, {wbase}, 0xc0")
            e("s_sub_u32 {off}, {off}, 0xc0")
            e("s_add_u32 {t0}, {wbase}, 0xc0")
            e("v_add_u32 {VT0}, {t0}, {VL4}")
            e("buffer_load_dword {win_next}, {VT0}, {in_rsrc}, 0 offen")
            e("s_branch " + L("top_slid"))

            lab("Oprev_fetch")                                # lzb.last_or(0) when the previous byte is not at hand
            e("s_mov_b32 {prev}, 0")
            e("s_cmp_eq_u32 {len}, 0")
            e("s_cbranch_scc1 " + L("lit_q"))
            e("s_add_u32 {t0}, {dict_base}, {len}")
            e("s_add_u32 {t0}, {t0}, -1")
            e("v_mov_b32 {VT0}, {t0}")
            e("buffer_load_ubyte {VT0}, {VT0}, {out_rsrc}, 0 offen")
            e("s_waitcnt vmcnt(0)")
            e("v_readfirstlane_b32 {prev}, {VT0}")
            e("s_branch " + L("lit_q"))

            lab("Omb_fetch")                                  # lzb.last_n(rep0 + 1)
            e("s_add_u32 {t1}, {dict_base}, {len}")
            e("s_sub_u32 {t1}, {t1}, {t0}")
            e("v_mov_b32 {VT0}, {t1}")
            e("buffer_load_ubyte {VT0}, {VT0}, {out_rsrc}, 0 offen")
            e("s_waitcnt vmcnt(0)")
            e("v_readfirstlane_b32 {mb}, {VT0}")
            e("s_branch " + L("lm_a"))

            lab("Orow_swap")                                  # park the cached literal row, unpack the new one
            e("v_lshl_or_b32 {VT0}, {u1}, 16, {u0}")
            e("v_lshl_or_b32 {VT1}, {u3}, 16, {u2}")
            e("s_lshl_b32 {t0}, {cur_row}, 1")
            e("s_set_gpr_idx_on {t0}, gpr_idx(DST)")
            e("v_mov_b32 " + LIT0 + ", {VT0}")
            e("v_mov_b32 " + LIT1 + ", {VT1}")
            e("s_set_gpr_idx_off")
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
            e("s_branch " + L("lit_r"))

            for name, code in [("Xdone_size", "DONE_SIZE"), ("Xdone_fin", "DONE_FIN"), ("Xeof", "INPUT_EOF"),
                               ("Xmarker", "MARKER"), ("Xlimit", "LIMIT"), ("Xlz_slow", "LZ_SLOW"),
                               ("Xmatch_dist_dict", "MATCH_DIST_DICT"), ("Xmatch_dist_out", "MATCH_DIST_OUT"),
                               ("Xlz_dist_dict", "LZ_DIST_DICT"), ("Xlz_dist_out", "LZ_DIST_OUT")]:
                lab(name)
                self.exit_with(code)

    def finish(self):
        # common exit: complete the pending match so that the C++ side sees memory and prev/mb up to date
        e, lab, L = self.e, self.lab, self.L
        lab("finish")
        e("s_cmp_eq_u32 {pend_n}, 0")
        e("s_cbranch_scc1 " + L("finish2"))
        self.finish_pending()
        lab("finish2")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")


def main():
    g = Gen()
    g.build()
    lines = g.main + g.cold + g.stubs
    g.cur = lines
    g.finish()
    out = []
    out.append("// GENERATED by tools/gen_fast_loop.py -- do not edit; edit the generator and re-run it.")
    out.append("// The symbol loop of decode_fast_asm_kernel as one inline-asm statement (see the generator's docstring).")
    out.append("// clang-format off")
    for k, v in EXIT.items():
        out.append("#define MILZMA_LOOP_EXIT_%s %du" % (k, v))
    out.append("#define MILZMA_FAST_LOOP_TEXT \\")
    for l in lines:
        out.append('  "%s\\n\\t" \\' % l.strip() if not l.endswith(":") else '  "%s\\n\\t" \\' % l)
    out.append('  ""')
    fixed = ['"+{v%d}"(d.lit[%d])' % (64 + i, i) for i in range(16)] + ['"+{v%d}"(d.posslot[%d])' % (80 + i, i) for i in range(4)]
    outs = ['[%s] "+s"(d.%s)' % (n, n) for n in OPS_INOUT_S] + ['[%s] "+v"(d.%s)' % (n, n) for n in OPS_INOUT_V] + fixed
    ins = ['[%s] "s"(d.%s)' % (n, n) for n in OPS_IN_S] + ['[%s] "v"(d.%s)' % (n, n) for n in OPS_IN_V]
    clob = sorted(set(S.values()), key=lambda r: int(r[1:])) + sorted(set(V.values()), key=lambda r: int(r[1:]))
    out.append("#define MILZMA_FAST_LOOP_OUTPUTS \\")
    out.append("  " + ", \\\n  ".join(outs))
    out.append("#define MILZMA_FAST_LOOP_INPUTS \\")
    out.append("  " + ", \\\n  ".join(ins))
    out.append("#define MILZMA_FAST_LOOP_CLOBBERS \\")
    out.append("  " + ", ".join('"%s"' % c for c in clob) + ', "vcc", "scc", "memory"')
    out.append("#define MILZMA_FAST_LOOP_UNIFORM(d, rf) \\")
    out.append("  " + " \\\n  ".join("d.%s = rf(d.%s);" % (n, n) for n in OPS_INOUT_S))
    out.append("// clang-format on")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lzma_rs_amd", "csrc", "fast_loop_asm.inc")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    n_ins = sum(1 for l in lines if not l.endswith(":"))
    print("wrote %s: %d instructions (%d in the main sequence)" % (os.path.normpath(path), n_ins,
                                                                    sum(1 for l in g.main if not l.endswith(":"))))


if __name__ == "__main__":
    main()
