"""Symbol-level LZMA encoder used ONLY to craft test vectors.

Written from the LZMA format description (range coder with 11-bit adaptive
probabilities, the 12-state literal/match/rep machine) so that tests can emit
arbitrary symbol sequences -- including ones no real compressor produces
(distances beyond the dictionary, markers followed by garbage, every lc/lp/pb)
-- and so that the reference's literal-only "dumb" encoder
(src/encode/dumbencoder.rs, used by its round-trip and option-matrix tests,
tests/lzma.rs:237-303) can be reproduced as a sequence of `lit` symbols.

Symbols:
  ("lit", byte)            literal
  ("match", length, dist)  new distance, dist >= 1 (1 = previous byte), 2 <= length <= 273
  ("rep", idx, length)     repeat distance idx in 0..3 ("long rep" when idx == 0)
  ("shortrep",)            rep0, length 1
  ("marker",)              end-of-stream marker (match with rep0 == 0xFFFFFFFF, len 2)
"""
import struct


class RangeEncoder:
    def __init__(self):
        self.low = 0
        self.range = 0xFFFFFFFF
        self.cache = 0
        self.cache_size = 1
        self.out = bytearray()

    def _shift_low(self):
        if self.low < 0xFF000000 or self.low > 0xFFFFFFFF:
            carry = self.low >> 32
            tmp = self.cache
            while True:
                self.out.append((tmp + carry) & 0xFF)
                tmp = 0xFF
                self.cache_size -= 1
                if self.cache_size == 0:
                    break
            self.cache = (self.low >> 24) & 0xFF
        self.cache_size += 1
        self.low = (self.low << 8) & 0xFFFFFFFF

    def encode_bit(self, probs, idx, bit):
        p = probs[idx]
        bound = (self.range >> 11) * p
        if bit == 0:
            self.range = bound
            probs[idx] = p + ((0x800 - p) >> 5)
        else:
            self.low += bound
            self.range -= bound
            probs[idx] = p - (p >> 5)
        while self.range < 0x01000000:
            self.range = (self.range << 8) & 0xFFFFFFFF
            self._shift_low()

    def encode_direct(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.range >>= 1
            if (value >> i) & 1:
                self.low += self.range
            while self.range < 0x01000000:
                self.range = (self.range << 8) & 0xFFFFFFFF
                self._shift_low()

    def finish(self):
        for _ in range(5):
            self._shift_low()
        return bytes(self.out)


def _tree(n):
    return [0x400] * n


class LenCoder:
    def __init__(self):
        self.choice = [0x400, 0x400]
        self.low = [_tree(8) for _ in range(16)]
        self.mid = [_tree(8) for _ in range(16)]
        self.high = _tree(256)

    def encode(self, rc, length, pos_state):
        """length is the stored value (real length - 2), 0..271."""
        if length < 8:
            rc.encode_bit(self.choice, 0, 0)
            _bittree(rc, self.low[pos_state], 3, length)
        elif length < 16:
            rc.encode_bit(self.choice, 0, 1)
            rc.encode_bit(self.choice, 1, 0)
            _bittree(rc, self.mid[pos_state], 3, length - 8)
        else:
            rc.encode_bit(self.choice, 0, 1)
            rc.encode_bit(self.choice, 1, 1)
            _bittree(rc, self.high, 8, length - 16)


def _bittree(rc, probs, nbits, value):
    m = 1
    for i in range(nbits - 1, -1, -1):
        b = (value >> i) & 1
        rc.encode_bit(probs, m, b)
        m = (m << 1) | b


def _bittree_reverse(rc, probs, nbits, value, offset=0):
    m = 1
    for _ in range(nbits):
        b = value & 1
        value >>= 1
        rc.encode_bit(probs, offset + m, b)
        m = (m << 1) | b


class LzmaSymbolEncoder:
    """Encodes a symbol list into a raw LZMA payload (no 13-byte header)."""

    def __init__(self, lc=3, lp=0, pb=2, history=b""):
        self.lc, self.lp, self.pb = lc, lp, pb
        self.rc = RangeEncoder()
        self.literal = [_tree(0x300) for _ in range(1 << (lc + lp))]
        self.pos_slot = [_tree(64) for _ in range(4)]
        self.align = _tree(16)
        self.pos_decoders = _tree(115)
        self.is_match = _tree(192)
        self.is_rep = _tree(12)
        self.is_rep_g0 = _tree(12)
        self.is_rep_g1 = _tree(12)
        self.is_rep_g2 = _tree(12)
        self.is_rep_0long = _tree(192)
        self.len_coder = LenCoder()
        self.rep_len_coder = LenCoder()
        self.state = 0
        self.rep = [0, 0, 0, 0]
        # `out` models the decoder's window; total_len drives pos_state/lit_state
        self.out = bytearray(history)
        self.total_len = len(history)

    # -- helpers -----------------------------------------------------------
    def _byte_back(self, dist):
        """Byte `dist` back in the window; 0 if it does not exist (crafted errors)."""
        if dist <= len(self.out):
            return self.out[len(self.out) - dist]
        return 0

    def _copy(self, length, dist):
        for _ in range(length):
            self.out.append(self._byte_back(dist))
        self.total_len += length

    def _pos_state(self):
        return self.total_len & ((1 << self.pb) - 1)

    # -- symbols -----------------------------------------------------------
    def lit(self, byte):
        rc = self.rc
        ps = self._pos_state()
        rc.encode_bit(self.is_match, (self.state << 4) + ps, 0)
        prev = self.out[-1] if self.out else 0
        lit_state = ((self.total_len & ((1 << self.lp) - 1)) << self.lc) + (prev >> (8 - self.lc))
        probs = self.literal[lit_state]
        result = 1
        if self.state >= 7:
            match_byte = self._byte_back(self.rep[0] + 1)
            while result < 0x100:
                match_bit = (match_byte >> 7) & 1
                match_byte = (match_byte << 1) & 0xFF
                bit = (byte >> (7 - (result.bit_length() - 1))) & 1
                rc.encode_bit(probs, ((1 + match_bit) << 8) + result, bit)
                result = (result << 1) | bit
                if match_bit != bit:
                    break
        while result < 0x100:
            bit = (byte >> (7 - (result.bit_length() - 1))) & 1
            rc.encode_bit(probs, result, bit)
            result = (result << 1) | bit
        self.out.append(byte)
        self.total_len += 1
        s = self.state
        self.state = 0 if s < 4 else (s - 3 if s < 10 else s - 6)

    def _encode_distance(self, rep0, stored_len):
        rc = self.rc
        len_state = min(stored_len, 3)
        if rep0 < 4:
            pos_slot = rep0
        else:
            n = rep0.bit_length()  # highest set bit index + 1
            pos_slot = ((n - 1) << 1) | ((rep0 >> (n - 2)) & 1)
        _bittree(rc, self.pos_slot[len_state], 6, pos_slot)
        if pos_slot >= 4:
            ndb = (pos_slot >> 1) - 1
            base = (2 | (pos_slot & 1)) << ndb
            rem = rep0 - base
            if pos_slot < 14:
                _bittree_reverse(rc, self.pos_decoders, ndb, rem, base - pos_slot)
            else:
                rc.encode_direct(rem >> 4, ndb - 4)
                _bittree_reverse(rc, self.align, 4, rem & 0xF)

    def match(self, length, dist):
        assert 2 <= length <= 273 and 1 <= dist <= 0xFFFFFFFF
        rc = self.rc
        ps = self._pos_state()
        rc.encode_bit(self.is_match, (self.state << 4) + ps, 1)
        rc.encode_bit(self.is_rep, self.state, 0)
        self.rep = [dist - 1, self.rep[0], self.rep[1], self.rep[2]]
        self.len_coder.encode(rc, length - 2, ps)
        self.state = 7 if self.state < 7 else 10
        self._encode_distance(dist - 1, length - 2)
        self._copy(length, dist)

    def marker(self):
        rc = self.rc
        ps = self._pos_state()
        rc.encode_bit(self.is_match, (self.state << 4) + ps, 1)
        rc.encode_bit(self.is_rep, self.state, 0)
        self.rep = [0xFFFFFFFF, self.rep[0], self.rep[1], self.rep[2]]
        self.len_coder.encode(rc, 0, ps)
        self.state = 7 if self.state < 7 else 10
        self._encode_distance(0xFFFFFFFF, 0)

    def shortrep(self):
        rc = self.rc
        ps = self._pos_state()
        rc.encode_bit(self.is_match, (self.state << 4) + ps, 1)
        rc.encode_bit(self.is_rep, self.state, 1)
        rc.encode_bit(self.is_rep_g0, self.state, 0)
        rc.encode_bit(self.is_rep_0long, (self.state << 4) + ps, 0)
        self.state = 9 if self.state < 7 else 11
        self._copy(1, self.rep[0] + 1)

    def rep_match(self, idx, length):
        assert 2 <= length <= 273 and 0 <= idx <= 3
        rc = self.rc
        ps = self._pos_state()
        rc.encode_bit(self.is_match, (self.state << 4) + ps, 1)
        rc.encode_bit(self.is_rep, self.state, 1)
        if idx == 0:
            rc.encode_bit(self.is_rep_g0, self.state, 0)
            rc.encode_bit(self.is_rep_0long, (self.state << 4) + ps, 1)
        else:
            rc.encode_bit(self.is_rep_g0, self.state, 1)
            if idx == 1:
                rc.encode_bit(self.is_rep_g1, self.state, 0)
            else:
                rc.encode_bit(self.is_rep_g1, self.state, 1)
                rc.encode_bit(self.is_rep_g2, self.state, idx - 2)
            d = self.rep[idx]
            for i in range(idx, 0, -1):
                self.rep[i] = self.rep[i - 1]
            self.rep[0] = d
        self.rep_len_coder.encode(rc, length - 2, ps)
        self.state = 8 if self.state < 7 else 11
        self._copy(length, self.rep[0] + 1)

    def encode(self, symbols):
        for s in symbols:
            kind = s[0]
            if kind == "lit":
                self.lit(s[1])
            elif kind == "match":
                self.match(s[1], s[2])
            elif kind == "rep":
                self.rep_match(s[1], s[2])
            elif kind == "shortrep":
                self.shortrep()
            elif kind == "marker":
                self.marker()
            else:
                raise ValueError(kind)
        return self

    def finish(self):
        return self.rc.finish()

    # -- LZMA2 chunking: the model survives, the range coder restarts -------
    def take_chunk(self):
        """Flush the range coder and start a new one (next LZMA2 chunk, no state reset)."""
        payload = self.rc.finish()
        self.rc = RangeEncoder()
        return payload

    def stored(self, data, reset_dict):
        """Mirror a stored LZMA2 chunk: state/reps untouched, window optionally reset."""
        if reset_dict:
            self.out = bytearray()
            self.total_len = 0
        self.out += data
        self.total_len += len(data)

    def reset_state(self, lc, lp, pb):
        """Mirror an LZMA2 state reset (optionally with new props); the window is kept."""
        out, total = self.out, self.total_len
        self.__init__(lc, lp, pb)
        self.out, self.total_len = out, total


def props_byte(lc, lp, pb):
    return lc + 9 * (lp + 5 * pb)


def lzma_header(lc=3, lp=0, pb=2, dict_size=0x800000, unpacked_size=None, write_size=True):
    """13-byte .lzma header (5 bytes when write_size is False)."""
    h = bytes([props_byte(lc, lp, pb)]) + struct.pack("<I", dict_size)
    if write_size:
        h += struct.pack("<Q", 0xFFFFFFFFFFFFFFFF if unpacked_size is None else unpacked_size)
    return h


def encode_lzma(symbols, lc=3, lp=0, pb=2, dict_size=0x800000, unpacked_size=None,
                write_size=True):
    """Returns (complete .lzma stream, plaintext the symbols describe)."""
    enc = LzmaSymbolEncoder(lc, lp, pb).encode(symbols)
    return lzma_header(lc, lp, pb, dict_size, unpacked_size, write_size) + enc.finish(), bytes(enc.out)


def dumb_encode(data, unpacked_size="marker", write_size=True):
    """What the reference's literal-only encoder produces (src/encode/dumbencoder.rs):
    lc3/lp0/pb2, dict 0x800000, every byte a literal, optional end marker.
    unpacked_size: "marker" -> header 0xFF..FF + EOS marker; int -> written, no marker.
    write_size False -> SkipWritingToHeader (5-byte header, no marker)."""
    syms = [("lit", b) for b in data]
    if write_size and unpacked_size == "marker":
        syms.append(("marker",))
        return encode_lzma(syms, unpacked_size=None)[0]
    if not write_size:
        return encode_lzma(syms, write_size=False)[0]
    return encode_lzma(syms, unpacked_size=unpacked_size)[0]


# ---- LZMA2 / XZ framing helpers (for crafted containers) -------------------

def lzma2_lzma_chunk(payload, unpacked_len, control, props=None):
    """One LZMA chunk: control byte (0x80|reset<<5 class), sizes, optional props, payload."""
    assert 1 <= unpacked_len <= (1 << 21) and 1 <= len(payload) <= 0x10000
    u = unpacked_len - 1
    out = bytes([control | (u >> 16)]) + struct.pack(">H", u & 0xFFFF) + struct.pack(">H", len(payload) - 1)
    if props is not None:
        out += bytes([props])
    return out + payload


def lzma2_stored_chunk(data, reset_dict):
    assert 1 <= len(data) <= 0x10000
    return bytes([1 if reset_dict else 2]) + struct.pack(">H", len(data) - 1) + data


def lz_parse(data, dict_size=1 << 16, max_len=273, use_reps=True):
    """A greedy LZ77 parse of `data` into the symbols above (3-byte hash heads, newest candidate only; repeat distances
    and short reps are preferred when they apply): real match structure -- long and short matches, near and far
    distances, every rep index -- for property sets no real compressor emits (liblzma refuses lc + lp > 4)."""
    n = len(data)
    head = {}
    reps = [0, 0, 0, 0]          # distances (>= 1); 0 = unset
    syms = []
    i = 0

    def mlen(pos, dist, cap):
        k = 0
        while k < cap and data[pos + k] == data[pos + k - dist]:
            k += 1
        return k

    while i < n:
        cap = min(max_len, n - i)
        best_len, best = 0, None
        if use_reps and i > 0:
            for idx, d in enumerate(reps):
                if d and d <= i:
                    k = mlen(i, d, cap)
                    if k >= 2 and k > best_len:
                        best_len, best = k, ("rep", idx, k)
            if best is None and reps[0] and reps[0] <= i and data[i] == data[i - reps[0]] and (len(syms) & 7) == 0:
                best_len, best = 1, ("shortrep",)
        if i + 3 <= n:
            key = data[i:i + 3]
            j = head.get(key)
            if j is not None and i - j <= dict_size:
                k = mlen(i, i - j, cap)
                if k >= 3 and k > best_len + 1:
                    best_len, best = k, ("match", k, i - j)
        if best is None:
            syms.append(("lit", data[i]))
            step = 1
        else:
            syms.append(best)
            step = best_len
            if best[0] == "match":
                reps = [best[2], reps[0], reps[1], reps[2]]
            elif best[0] == "rep" and best[1] > 0:
                d = reps[best[1]]
                for t in range(best[1], 0, -1):
                    reps[t] = reps[t - 1]
                reps[0] = d
        for p in range(i, min(i + step, n - 2)):
            head[data[p:p + 3]] = p
        i += step
    return syms
