"""A small CABAC ENCODER in Python (ITU-T H.264 9.3.4.2-9.3.4.5, written from
the standard's flow charts) for the subset of slice data the engine's CABAC
decoder accepts.  Test infrastructure: it is the independent implementation in
the OPPOSITE direction that the decoders (oracle/orc_cabac.c and
kernels/cabac_decode.hip) are checked against with values the reference's own
writers never produce (prediction modes, chroma modes, QP deltas)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_T = None


def tables():
    """rangeTabLPS, transIdxLPS/MPS and (m, n) from the shared table file."""
    global _T
    if _T is None:
        txt = open(os.path.join(ROOT, "libh264_amd", "csrc", "common",
                                "h264_cabac_tables.inc")).read()
        lps, nl, nm, mn = {}, {}, {}, {}
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        num = r"\(([-0-9,\s]*)\)"
        for m in re.finditer(r"H264_LPS" + num, txt):
            v = [int(x) for x in m.group(1).split(",")]
            lps[v[0]] = v[1:5]
        for m in re.finditer(r"H264_TRANS" + num, txt):
            v = [int(x) for x in m.group(1).split(",")]
            nl[v[0]], nm[v[0]] = v[1], v[2]
        for m in re.finditer(r"H264_MN" + num, txt):
            v = [int(x) for x in m.group(1).split(",")]
            mn[v[0]] = [(v[1], v[2]), (v[3], v[4]), (v[5], v[6]), (v[7], v[8])]
        assert len(lps) == 64 and len(nl) == 64 and len(mn) == 1024
        _T = (lps, nl, nm, mn)
    return _T


class Encoder:
    def __init__(self, slice_type, cabac_init_idc, qp):
        lps, nl, nm, mn = tables()
        self.lps, self.nl, self.nm = lps, nl, nm
        col = 0 if slice_type % 5 in (2, 4) else 1 + cabac_init_idc
        q = min(51, max(1, qp))  # the reference clips to 1..51
        self.st = []
        for c in range(1024):
            m, n = mn[c][col]
            idx = ((m * q) >> 4) + n
            if idx <= 63:
                self.st.append([63 - max(1, idx), 0])
            else:
                self.st.append([min(126, idx) - 64, 1])
        self.low, self.range = 0, 510
        self.first, self.outstanding = True, 0
        self.bits = []

    def _put(self, b):
        if self.first:
            self.first = False
        else:
            self.bits.append(b)
        self.bits.extend([1 - b] * self.outstanding)
        self.outstanding = 0

    def _renorm(self):
        while self.range < 256:
            if self.low < 256:
                self._put(0)
            elif self.low >= 512:
                self.low -= 512
                self._put(1)
            else:
                self.low -= 256
                self.outstanding += 1
            self.range <<= 1
            self.low <<= 1

    def decision(self, ci, b):
        s = self.st[ci]
        r_lps = self.lps[s[0]][(self.range >> 6) & 3]
        self.range -= r_lps
        if b != s[1]:
            self.low += self.range
            self.range = r_lps
            if s[0] == 0:
                s[1] ^= 1
            s[0] = self.nl[s[0]]
        else:
            s[0] = self.nm[s[0]]
        self._renorm()

    def terminate(self, b):
        self.range -= 2
        if b:
            self.low += self.range
            self.range = 2
            self._renorm()
            self._put((self.low >> 9) & 1)
            self.bits.append((self.low >> 8) & 1)
            self.bits.append(1)  # rbsp_stop_one_bit
        else:
            self._renorm()

    def payload(self):
        bits = list(self.bits)
        while len(bits) % 8:
            bits.append(0)
        return bytes(int("".join(map(str, bits[i:i + 8])), 2)
                     for i in range(0, len(bits), 8))


def encode_i16_slice(mbs, width, first_mb, cabac_init_idc, qp, cat=1):
    """mbs: list of (pred_mode 0..3, chroma_mode 0..3, qp_delta).  Returns the
    arithmetic-coded bytes (they start byte aligned)."""
    e = Encoder(2, cabac_init_idc, qp)
    info = {}
    prev_nz = 0
    for i, (pred, chroma, qpd) in enumerate(mbs):
        addr = first_mb + i
        a = info.get(addr - 1) if (addr % width != 0 and i >= 1) else None
        b = info.get(addr - width) if i >= width else None
        e.decision(3 + (a is not None) + (b is not None), 1)
        e.terminate(0)
        e.decision(3 + 3, 0)
        e.decision(3 + 4, 0)
        e.decision(3 + 6, pred >> 1)
        e.decision(3 + 7, pred & 1)
        if cat in (1, 2):
            inc = int(a is not None and a != 0) + int(b is not None and b != 0)
            e.decision(64 + inc, int(chroma > 0))
            if chroma > 0:
                e.decision(64 + 3, int(chroma > 1))
                if chroma > 1:
                    e.decision(64 + 3, int(chroma > 2))
        k = 2 * qpd - 1 if qpd > 0 else -2 * qpd
        e.decision(60 + prev_nz, int(k > 0))
        if k > 0:
            e.decision(60 + 2, int(k > 1))
            for j in range(2, k + 1):
                e.decision(60 + 3, int(j < k))
        prev_nz = int(qpd != 0)
        e.decision(85 + (a is None) + 2 * (b is None), 0)
        info[addr] = chroma
        e.terminate(int(i == len(mbs) - 1))
    return e.payload()


def encode_skip_slice(n, slice_type, cabac_init_idc, qp):
    e = Encoder(slice_type, cabac_init_idc, qp)
    off = 24 if slice_type % 5 == 1 else 11
    for i in range(n):
        e.decision(off, 1)  # neighbours are skipped: ctxIdxInc 0
        e.terminate(int(i == n - 1))
    return e.payload()


def escape_tail(head, tail):
    """head: finished NAL bytes (already escaped); tail: raw RBSP bytes to
    append with emulation prevention continuing from head."""
    out = bytearray(head)
    zeros = 0
    for b in head[-4:]:
        zeros = zeros + 1 if b == 0 else 0
    for b in tail:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)
