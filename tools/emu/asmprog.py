#!/usr/bin/env python3
"""Front end of the asm-loop emulator (tools/emu/emu.cpp): turns the text tools/gen_fast_loop.py
generates into the emulator's pre-parsed program and drives one wavefront through a raw LZMA stream
the way decode_fast_asm_kernel / AsmDecoder::process (lzma_rs_amd/csrc/decode_fast_asm.hip.h) do.

Test / tuning infrastructure only (tests/test_asm_emulator.py, tools/emu/profile.py): it lets a
generator change be checked bit-exactly against the oracle on the CPU, and gives exact executed-
instruction counts per class.  Nothing in the product imports it."""
import ctypes
import os
import re
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))

K_S, K_V, K_I, K_L, K_VCC = 0, 1, 2, 3, 4
GPR_IDX = {"SRC0": 1, "SRC1": 2, "SRC2": 4, "DST": 8}

ST_OK, ST_INPUT_EOF, ST_RC_INIT = "OK", "INPUT_EOF", "RC_INIT"


def _lib():
    so = os.path.join(HERE, "libasmemu.so")
    src = os.path.join(HERE, "emu.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", src, "-o", so])
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.emu_create.restype = vp
    L.emu_create.argtypes = [ctypes.c_char_p]
    L.emu_destroy.argtypes = [vp]
    L.emu_error.restype = ctypes.c_char_p
    L.emu_error.argtypes = [vp]
    L.emu_set_mem.argtypes = [vp, vp, ctypes.c_uint64]
    L.emu_set_s.argtypes = [vp, ctypes.c_int, ctypes.c_uint32]
    L.emu_get_s.restype = ctypes.c_uint32
    L.emu_get_s.argtypes = [vp, ctypes.c_int]
    L.emu_set_v.argtypes = [vp, ctypes.c_int, vp]
    L.emu_get_v.argtypes = [vp, ctypes.c_int, vp]
    L.emu_set_hwreg.argtypes = [vp, ctypes.c_uint32]
    L.emu_lds_write.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint32]
    L.emu_run.restype = ctypes.c_long
    L.emu_run.argtypes = [vp, ctypes.c_int, ctypes.c_long]
    L.emu_counts.argtypes = [vp, vp, vp]
    L.emu_reset_counts.argtypes = [vp]
    L.emu_size.restype = ctypes.c_int
    L.emu_size.argtypes = [vp]
    return L


def split_operands(text):
    """comma-separated operands; commas inside (...) or [...] do not split"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Program:
    """lines: the generator's instruction / label lines.  regmap: operand name -> 'sN' / 'vN'."""

    def __init__(self, lines, regmap):
        self.regmap = regmap
        self.labels = {}
        self.text = []    # instruction text per index
        self.region = []  # innermost preceding label per instruction
        self.tag = []     # (section, role) per instruction, from the generator (profile.py --sections)
        cur = "entry"
        for ln in lines:
            s = ln.strip()
            if not s:
                continue
            if s.startswith("."):      # an assembler directive (.p2align in front of a branch target): nothing to execute
                continue
            if s.endswith(":"):
                name = s[:-1]
                self.labels[name] = len(self.text)
                cur = name
                continue
            self.text.append(s)
            self.region.append(cur)
            self.tag.append((getattr(ln, "sec", ""), getattr(ln, "role", "")))
        self.encoded = [self._encode(i, t) for i, t in enumerate(self.text)]

    def _reg(self, tok):
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return (K_S, int(m.group(1)))
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            return (K_S, int(m.group(1)))
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return (K_V, int(m.group(1)))
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return (K_V, int(m.group(1)))
        if tok == "vcc":
            return (K_VCC, 0)
        if tok == "m0":
            return (K_S, 124)
        return None

    def _operand(self, tok):
        tok = tok.strip()
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        if m:
            tok = self.regmap[m.group(1)]
        r = self._reg(tok)
        if r:
            return r
        if re.fullmatch(r"-?\d+", tok):
            return (K_I, int(tok) & 0xFFFFFFFF)
        if re.fullmatch(r"0x[0-9a-fA-F]+", tok):
            return (K_I, int(tok, 16) & 0xFFFFFFFF)
        if re.fullmatch(r"-?\d+\.\d+", tok):
            return (K_I, struct.unpack("<I", struct.pack("<f", float(tok)))[0])
        m = re.fullmatch(r"gpr_idx\(([\w,\s]+)\)", tok)
        if m:
            return (K_I, sum(GPR_IDX[x.strip()] for x in m.group(1).split(",")))
        if tok.startswith("hwreg("):
            return (K_I, 0)
        if tok in self.labels:
            return (K_L, self.labels[tok])
        # label arithmetic: addresses are 4 * instruction index
        expr = re.sub(r"L\w+%=", lambda mm: str(4 * self.labels[mm.group(0)]), tok)
        if re.fullmatch(r"[\d\s()+\-*/]+", expr):
            return (K_I, int(eval(expr.replace("/", "//"))) & 0xFFFFFFFF)
        raise ValueError("cannot parse operand %r" % tok)

    def _encode(self, idx, text):
        parts = text.split(None, 1)
        op = parts[0]
        if op.endswith("_e64"):     # (explicit VOP3 encoding: same semantics)
            op = op[:-4]
        ops = split_operands(parts[1]) if len(parts) > 1 else []
        if op in ("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_set_gpr_idx_off"):
            ops = []
        if op.startswith("buffer_"):
            # vdata, vaddr, srsrc, "soffset offen [mods]"
            last = ops[3].split()
            assert "offen" in last, text
            ops = ops[:3] + [last[0]]
        if op == "s_load_dword":
            ops = ops[:2] + [ops[2].split()[0]]
        if op.startswith("ds_"):
            ops = [o for o in ops if not o.startswith("offset:")] + [o.split(":")[1] for o in ops if o.startswith("offset:")]
        enc = [self._operand(o) for o in ops]
        return "%s %d %s" % (op, len(enc), " ".join("%d %d" % e for e in enc))

    def blob(self):
        return ("\n".join(self.encoded) + "\n").encode()


def classify(text):
    op = text.split()[0]
    if op in ("s_nop", "s_waitcnt", "s_sleep", "s_setprio"):
        return "misc"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_call_b64"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_"):
        return "vmem"
    return "other"


class AsmLoop:
    """One wavefront running the generated loop on a raw LZMA stream."""

    def __init__(self, lp0=True, gen_module=None, pb4=False, hbm=False):
        import gen_fast_loop as G
        self.G = gen_module or G
        G = self.G
        self.hbm = hbm
        g = G.Gen(lp0, pb4, hbm=hbm)
        self.lit_regs, self.ps0 = G.LIT_REGS, int(G.PS0[1:])  # (this variant's fixed register numbering)
        g.build()
        lines = g.main + g.cold + getattr(g, "cold2", []) + g.stubs
        g.cur = lines
        g.finish()
        # operand -> register: scalars from s0 (4-aligned quads for the two descriptors), vectors from v0
        fixed = getattr(G, "FIXED_OPERANDS", {})
        self.regmap = dict(fixed)
        s_next = 0
        for name in G.OPS_INOUT_S + G.OPS_IN_S:
            if name in self.regmap:
                continue
            if name in ("in_rsrc", "out_rsrc", "lit_rsrc"):
                s_next = (s_next + 3) & ~3
                self.regmap[name] = "s[%d:%d]" % (s_next, s_next + 3)
                s_next += 4
            elif name == "flagptr":
                s_next = (s_next + 1) & ~1
                self.regmap[name] = "s[%d:%d]" % (s_next, s_next + 1)
                s_next += 2
            else:
                self.regmap[name] = "s%d" % s_next
                s_next += 1
        assert s_next <= 64, "operand SGPRs collide with the generator's fixed temporaries"
        v_next = 0
        for name in G.OPS_INOUT_V + G.OPS_IN_V:
            if name in self.regmap:
                continue
            self.regmap[name] = "v%d" % v_next
            v_next += 1
        assert v_next <= 64
        self.prog = Program(lines, self.regmap)
        self.L = _lib()
        self.h = self.L.emu_create(self.prog.blob())
        err = self.L.emu_error(self.h)
        if err:
            raise RuntimeError("emulator: " + err.decode())
        self.n = self.L.emu_size(self.h)

    def close(self):
        if self.h:
            self.L.emu_destroy(self.h)
            self.h = None

    # ---- register access by operand name -------------------------------------------------------------
    def _sidx(self, name):
        r = self.regmap[name]
        m = re.match(r"s\[?(\d+)", r)
        return int(m.group(1))

    def _vidx(self, name):
        return int(re.match(r"v\[?(\d+)", self.regmap[name]).group(1))

    def sset(self, name, val):
        self.L.emu_set_s(self.h, self._sidx(name), int(val) & 0xFFFFFFFF)

    def sget(self, name):
        return self.L.emu_get_s(self.h, self._sidx(name))

    def vset(self, idx, val):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(val, dtype=np.uint32), (64,)))
        self.L.emu_set_v(self.h, idx, a.ctypes.data)

    def vget(self, idx):
        a = np.zeros(64, dtype=np.uint32)
        self.L.emu_get_v(self.h, idx, a.ctypes.data)
        return a

    def set_rsrc(self, name, base, records):
        i = self._sidx(name)
        for k, w in enumerate((base & 0xFFFFFFFF, (base >> 32) & 0xFFFF, records, 0x00020000)):
            self.L.emu_set_s(self.h, i + k, w)


    def _reset_model(self, lane):
        """reset_model: every probability 0x400 (registers and the LDS rows), the row caches empty"""
        G = self.G
        for name in G.OPS_INOUT_V:
            if name.startswith("m_") or name in ("u0", "u1", "u2", "u3"):
                self.vset(self._vidx(name), 0x400)
        for i in range(self.lit_regs):
            self.vset(getattr(G, "VBASE", 64) + i, 0x04000400)
        for i in range(4):
            self.vset(self.ps0 + i, 0x400)
        lds = np.full(self.lit_regs // 2 * 64 * 4, 0x04000400, dtype=np.uint32)
        self.L.emu_lds_write(self.h, 0, lds.ctypes.data, lds.size * 4)
        self.vset(self._vidx("v_lane"), lane)
        self.vset(self._vidx("pend_val"), 0)
        if "vtag" in self.regmap:   # (the walked row 0 owns its slot; everything else empty)
            self.vset(self._vidx("vtag"), np.where(lane == 0, 0, 0xFFFFFFFF).astype(np.uint32))
            self.vset(self._vidx("vtagm"), 0xFFFFFFFF)

    # ---- an LZMA2 unit: the packet walk of decode_unit (lzma2.rs:52-229) around the same loop -- well-formed streams of the fast class only
    #      (lc + lp <= 3; what bench.py's .xz recipe holds): the instruction mix of configs[3], not an error-path mirror -------------------------
    def decode_lzma2(self, data, out_cap, max_steps=1 << 40):
        """Returns dict(status, out, executed, chunks).  Stored chunks are copied by the front end (the kernel's C++ does that); every LZMA
        chunk runs in the loop with a fresh range coder, the model / state reset as its control byte says."""
        G = self.G
        data = bytes(data)
        IN0 = 64
        in_span = (len(data) + 63 + 128) & ~63
        OUT0 = IN0 + in_span + 64
        mem = np.zeros(OUT0 + out_cap + 1024, dtype=np.uint8)
        mem[IN0:IN0 + len(data)] = np.frombuffer(data, dtype=np.uint8)
        mem[0] = 1
        if "flagptr" in self.regmap:
            i = self._sidx("flagptr")
            self.L.emu_set_s(self.h, i, 0)
            self.L.emu_set_s(self.h, i + 1, 0)
        self._mem = mem
        self.L.emu_set_mem(self.h, mem.ctypes.data, mem.size)
        self.L.emu_set_hwreg(self.h, 0)
        lane = np.arange(64, dtype=np.uint32)
        S = self.sset
        self._reset_model(lane)
        S("len", 0)
        S("state", 0)
        for r in ("rep0", "rep1", "rep2", "rep3"):
            S(r, 0)
        S("cur_row", 0)
        S("mlen", 0)
        for p in ("prof_wm", "prof_nm", "prof_wc", "prof_nc"):
            S(p, 0)
        S("ldsbase", 0)
        S("qtop", 0xFFFFFFFF)
        S("known", 1)
        S("dict_size", 1 << 26)            # (the XZ filter's dictionary bounds distances; well-formed input: never reached)
        S("out_lim", out_cap)
        S("safe_len", out_cap - 273 if out_cap >= 273 else 0)
        self.set_rsrc("out_rsrc", OUT0, out_cap)
        pos, executed, chunks, have_props = 0, 0, 0, False
        while True:
            c = data[pos]
            if c == 0:
                return dict(status=ST_OK, out=mem[OUT0:OUT0 + self.sget("len")].tobytes(), executed=executed, chunks=chunks, in_consumed=pos + 1)
            if c < 0x80:                   # stored chunk (1: dictionary reset, 2: none)
                n = ((data[pos + 1] << 8) | data[pos + 2]) + 1
                ln = self.sget("len")
                mem[OUT0 + ln:OUT0 + ln + n] = np.frombuffer(data[pos + 3:pos + 3 + n], dtype=np.uint8)
                S("len", ln + n)
                pos += 3 + n
                continue
            unpacked = (((c & 0x1F) << 16) | (data[pos + 1] << 8) | data[pos + 2]) + 1
            packed = ((data[pos + 3] << 8) | data[pos + 4]) + 1
            hdr = 5
            if c >= 0xC0:
                props = data[pos + 5]
                hdr = 6
                lc, lp, pb = props % 9, (props // 9) % 5, props // 45
                assert lc + lp <= 3 and pb <= 2, "the front end mirrors the LP0 / GEN variants only"
                S("lc", lc)
                S("lc8", 8 - lc)
                S("lpmask", (1 << lp) - 1)
                S("pbmask", (1 << pb) - 1)
                have_props = True
            assert have_props
            if c >= 0xA0:                  # state reset (with or without new properties)
                self._reset_model(lane)
                S("state", 0)
                for r in ("rep0", "rep1", "rep2", "rep3"):
                    S(r, 0)
                S("cur_row", 0)
            payload = data[pos + hdr:pos + hdr + packed]
            base = IN0 + pos + hdr

            def window(wpos, payload=payload):
                w = np.zeros(64, dtype=np.uint32)
                seg = payload[wpos:wpos + 64]
                w[:len(seg)] = np.frombuffer(seg, dtype=np.uint8) if seg else 0
                return w

            self.set_rsrc("in_rsrc", base, packed)
            self.vset(self._vidx("winb"), window(0))
            self.vset(self._vidx("winb_next"), window(64))
            S("range", 0xFFFFFFFF)
            S("code", int.from_bytes(payload[1:5], "big"))
            S("off", 5)
            S("lim", packed)
            S("wbase", 0)
            S("prev", 0)
            S("mb", 0xFFFFFFFF)
            S("pend_n", G.PEND_UNKNOWN if self.sget("len") else 0)     # (the previous byte is fetched from the output by the loop's entry)
            S("pend_pos", 0)
            S("exitcode", 0)
            S("tbl_ready", 0)
            S("target", self.sget("len") + unpacked)
            while True:
                n = self.L.emu_run(self.h, 0, max_steps)
                if n < 0:
                    raise RuntimeError("emulator: " + self.L.emu_error(self.h).decode())
                executed += n
                ex = self.sget("exitcode") & 0xFF
                S("exitcode", ex)
                if ex != G.EXIT["LZ_SLOW"]:
                    break
                ln, mlen, dist = self.sget("len"), self.sget("mlen"), self.sget("rep0") + 1
                for i in range(mlen):
                    mem[OUT0 + ln + i] = mem[OUT0 + ln - dist + i]
                S("pend_n", G.PEND_UNKNOWN)
                S("mb", 0xFFFFFFFF)
                S("len", ln + mlen)
            if ex != G.EXIT["DONE_SIZE"]:
                name = {v: k for k, v in G.EXIT.items()}[ex]
                return dict(status=name, out=mem[OUT0:OUT0 + self.sget("len")].tobytes(), executed=executed, chunks=chunks, in_consumed=pos)
            chunks += 1
            pos += hdr + packed

    # ---- the kernel around the loop (decode_fast_asm.hip.h), raw LZMA units only ------------------------------
    def decode_raw(self, payload, lc, lp, pb, dict_size, unpacked_size, out_cap=None, max_steps=1 << 40, hw_slot=0, quantum=None, feed_views=None,
                   partial=False):
        """Returns dict(status, out, len, in_consumed, executed).  unpacked_size None = unknown (marker mode).
        quantum: output bytes after which the loop yields at the next symbol top (the time-sliced launches); the front end then does
        what the kernel's resume does (re-seeks the reader from its position, has the per-lane tables rebuilt) and re-enters.
        feed_views: ascending prefix lengths of `payload` -- the input arrives in VIEWS (MILZMA_DECODE_FEED): the loop runs with the FEED bit
        on the first feed_views[0] bytes, leaves with NEED_INPUT at a symbol top near the view's end, and is re-entered on the next, longer
        view (the reader re-seeked at its position, as the kernel's resume does); the whole payload comes last, without the bit.
        The result then carries `feeds` = [(view length, reader position at the stop), ...].
        partial (with feed_views): the crate's Partial mode at an END MARKER (MILZMA_KIND_PARTIAL, lzma.rs:493-495 / :507-509) the way the
        time-sliced kernel does it: a view's tail (fewer than 20 bytes left) is decoded on as if the view were the last (the trial pass;
        here without the roll-back: every symbol of the tail must be complete), and a marker that ends the view with code == 0 does not
        end the unit -- the loop is re-entered on the next view from the marker's state."""
        G = self.G
        if out_cap is None:
            out_cap = unpacked_size if unpacked_size is not None else len(payload) * 64 + 4096
        views = list(feed_views or []) + [len(payload)]
        full_payload = payload
        payload = full_payload[:views[0]]
        feeds = []
        in_len = len(payload)
        IN0 = 64  # (bytes 0..3 of the emulated memory: the "last block has started" flag, set)
        in_span = (len(full_payload) + 63 + 128) & ~63
        OUT0 = IN0 + in_span + 64
        SLAB0 = (OUT0 + out_cap + 512 + 255) & ~255
        slab_bytes = (0x600 << (lc + lp)) if self.hbm else 0
        mem = np.zeros(SLAB0 + slab_bytes + 64, dtype=np.uint8)
        if slab_bytes:     # every probability 0x400 (the host memsets the slab before the launch)
            mem[SLAB0:SLAB0 + slab_bytes].view(np.uint16)[:] = 0x400
        mem[IN0:IN0 + len(full_payload)] = np.frombuffer(bytes(full_payload), dtype=np.uint8)
        mem[0] = 1
        if "flagptr" in self.regmap:
            i = self._sidx("flagptr")
            self.L.emu_set_s(self.h, i, 0)
            self.L.emu_set_s(self.h, i + 1, 0)
        self._mem = mem
        self.L.emu_set_mem(self.h, mem.ctypes.data, mem.size)
        self.L.emu_set_hwreg(self.h, hw_slot)
        lane = np.arange(64, dtype=np.uint32)

        def window(wpos):
            w = np.zeros(64, dtype=np.uint32)
            for l in range(64):
                if 0 <= wpos + l < in_len:
                    w[l] = payload[wpos + l]
            return w

        self._reset_model(lane)
        if "lit_rsrc" in self.regmap:
            self.set_rsrc("lit_rsrc", SLAB0, slab_bytes)
        # reader: seek(0, in_len), then rc_init
        wbase, off, lim = 0, 0, in_len
        if in_len < 5:
            return dict(status=ST_RC_INIT, out=b"", len=0, in_consumed=in_len, executed=0)
        code = int.from_bytes(bytes(payload[1:5]), "big")
        off = 5
        self.vset(self._vidx("winb"), window(0))
        self.vset(self._vidx("winb_next"), window(64))
        S = self.sset
        S("range", 0xFFFFFFFF)
        S("code", code)
        S("off", off)
        S("lim", lim)
        S("wbase", wbase)
        S("len", 0)
        S("state", 0)
        for r in ("rep0", "rep1", "rep2", "rep3"):
            S(r, 0)
        S("prev", 0)
        S("mb", 0xFFFFFFFF)
        S("pend_n", 0)
        S("pend_pos", 0)
        S("cur_row", 0)
        S("mlen", 0)
        S("exitcode", 0)
        for p in ("prof_wm", "prof_nm", "prof_wc", "prof_nc"):
            S(p, 0)
        known = unpacked_size is not None
        clamped = known and unpacked_size > 0xFFFFFFFF
        S("known", 1 if known else 0)
        S("target", unpacked_size if (known and not clamped) else 0xFFFFFFFF)
        out_lim = out_cap
        S("out_lim", out_lim)
        S("safe_len", out_lim - 273 if out_lim >= 273 else 0)
        S("dict_size", dict_size)
        S("lc", lc)
        S("lc8", (8 - lc) | ((1 << G.FEED_BIT) if len(views) > 1 else 0))
        S("lpmask", (1 << lp) - 1)
        S("pbmask", (1 << pb) - 1)
        S("ldsbase", 0)
        self.set_rsrc("in_rsrc", IN0, in_len)
        self.set_rsrc("out_rsrc", OUT0, out_cap)
        executed = 0
        status = None
        yields = 0
        trial = False
        while True:
            S("qtop", min(0xFFFFFFFF, self.sget("len") + quantum) if quantum else 0xFFFFFFFF)
            n = self.L.emu_run(self.h, 0, max_steps)
            if n < 0:
                raise RuntimeError("emulator: " + self.L.emu_error(self.h).decode())
            executed += n
            ex = self.sget("exitcode") & 0xFF      # (bit 8: "re-seek before reading on" -- positions are right either way)
            S("exitcode", ex)
            marker_park = False
            if partial and ex == G.EXIT["MARKER"] and len(views) > 1 and self.sget("state") in (7, 10):
                rem = (self.sget("lim") - self.sget("off")) & 0xFFFFFFFF
                marker_park = rem == 0 and self.sget("code") == 0      # AsmDecoder::process: parked behind the marker
                if marker_park:
                    S("mlen", 0)
            if partial and ex == G.EXIT["NEED_INPUT"] and not trial:
                # the view's tail: on as if the view were the last (no FEED bit), from the registers as they are
                trial = True
                v = (self.sget("wbase") + self.sget("off")) & 0xFFFFFFFF
                r = in_len - v
                S("lc8", 8 - lc)
                S("wbase", v & ~63)
                S("off", v & 63)
                S("lim", (v & 63) + r)
                self.vset(self._vidx("winb"), window(v & ~63))
                self.vset(self._vidx("winb_next"), window((v & ~63) + 64))
                S("tbl_ready", 0)
                continue
            if ex == G.EXIT["NEED_INPUT"] or marker_park:
                # MILZMA_DECODE_FEED: parked at a symbol top near the end of the view; the next view is longer (here: of the same buffer)
                v = (self.sget("wbase") + self.sget("off")) & 0xFFFFFFFF
                feeds.append((in_len, v))
                assert len(views) > 1 and in_len - v < G.FEED_MARGIN, (in_len, v)
                trial = False
                views.pop(0)
                payload = full_payload[:views[0]]
                in_len = len(payload)
                self.set_rsrc("in_rsrc", IN0, in_len)
                S("lc8", (8 - lc) | ((1 << G.FEED_BIT) if len(views) > 1 else 0))
                S("wbase", v & ~63)
                S("off", v & 63)
                S("lim", (v & 63) + (in_len - v))
                self.vset(self._vidx("winb"), window(v & ~63))
                self.vset(self._vidx("winb_next"), window((v & ~63) + 64))
                S("tbl_ready", 0)
                continue
            if ex == G.EXIT["QUANTUM"]:
                # resume: seek(vpos(), rem()) -- aligned windows, off = lane -- and tables rebuilt on entry
                yields += 1
                v = (self.sget("wbase") + self.sget("off")) & 0xFFFFFFFF
                r = (self.sget("lim") - self.sget("off")) & 0xFFFFFFFF
                S("wbase", v & ~63)
                S("off", v & 63)
                S("lim", (v & 63) + r)
                self.vset(self._vidx("winb"), window(v & ~63))
                self.vset(self._vidx("winb_next"), window((v & ~63) + 64))
                S("tbl_ready", 0)
                continue
            if ex != G.EXIT["LZ_SLOW"]:
                break
            # append_lz_slow: matches of >= 64 bytes or running into the output limit
            ln, mlen, dist = self.sget("len"), self.sget("mlen"), self.sget("rep0") + 1
            nb = mlen
            clipped = False
            if ln + mlen > out_lim:
                nb = max(0, out_lim - ln)
                clipped = True
            for i in range(nb):
                mem[OUT0 + ln + i] = mem[OUT0 + ln - dist + i]
            S("pend_n", G.PEND_UNKNOWN)
            S("mb", 0xFFFFFFFF)
            if clipped:
                if known and out_lim >= unpacked_size:
                    S("len", ln + mlen)
                    continue
                S("len", ln + nb)
                status = "OUT_FULL"
                break
            S("len", ln + mlen)
        ln = self.sget("len")
        if status is None:
            name = {v: k for k, v in G.EXIT.items()}[ex]
            if name == "DONE_SIZE":
                status = ST_OK if ln == unpacked_size else "SIZE_MISMATCH"
            elif name == "DONE_FIN":
                status = ST_OK
            elif name == "MARKER" and self.sget("state") not in (7, 10):
                # a REP match that finds the marker's distance in the history (only behind a marker, `partial`): append_lz's error
                status = "LZ_DIST_DICT"
            elif name == "MARKER":
                rem = (self.sget("lim") - self.sget("off")) & 0xFFFFFFFF
                status = ST_OK if (rem == 0 and self.sget("code") == 0) else "MARKER_TRAILING"
                if status == ST_OK and known and ln != unpacked_size:   # lzma.rs:513-521 applies after the marker too
                    status = "SIZE_MISMATCH"
            elif name == "LIMIT":
                status = "OUT_FULL"
            else:
                status = name
        in_consumed = (self.sget("wbase") + self.sget("off")) & 0xFFFFFFFF   # (an end-aligned last window may start "before" 0)
        if status == ST_INPUT_EOF:
            # the failing normalisation had already advanced its byte offset: AsmDecoder::process takes that one back
            in_consumed = (in_consumed - 1) & 0xFFFFFFFF
            assert in_consumed == in_len, (in_consumed, in_len)
        return dict(status=status, out=mem[OUT0:OUT0 + min(ln, out_cap)].tobytes(), len=ln, in_consumed=in_consumed,
                    executed=executed, yields=yields, feeds=feeds, rep0=self.sget("rep0"))

    # ---- executed-instruction statistics ------------------------------------------------------------------
    def counts(self):
        c = np.zeros(self.n, dtype=np.uint64)
        t = np.zeros(self.n, dtype=np.uint64)
        self.L.emu_counts(self.h, c.ctypes.data, t.ctypes.data)
        return c, t

    def reset_counts(self):
        self.L.emu_reset_counts(self.h)

    def mix(self):
        """executed instructions per class, and taken branches"""
        c, t = self.counts()
        out = {}
        for i, text in enumerate(self.prog.text):
            k = classify(text)
            out[k] = out.get(k, 0) + int(c[i])
        out["taken"] = int(t.sum())
        out["total"] = int(c.sum())
        return out
