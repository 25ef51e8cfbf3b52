"""Just enough of ISO/IEC 14496-12 / -15 to get the NAL units of the first AVC
track out of an MP4 file: sample table (stsz / stsc / stco / co64), the avcC
record with its parameter sets, length-prefixed NAL units per sample.  Test
infrastructure (tests/test_cabac_real_streams.py)."""
import struct
def _boxes(data, off, end):
    while off + 8 <= end:
        size, typ = struct.unpack('>I4s', data[off:off+8])
        hdr = 8
        if size == 1:
            size = struct.unpack('>Q', data[off+8:off+16])[0]; hdr = 16
        if size == 0: size = end - off
        yield typ, off + hdr, off + size
        off += size
def _find(data, off, end, path):
    for typ, a, b in _boxes(data, off, end):
        if typ == path[0]:
            if len(path) == 1:
                yield a, b
            else:
                yield from _find(data, a, b, path[1:])
def video_nals(path):
    """(sps list, pps list, [sample -> list of NAL units]) of the first avc1 track"""
    data = open(path, 'rb').read()
    for ta, tb in _find(data, 0, len(data), [b'moov', b'trak']):
        stbl = list(_find(data, ta, tb, [b'mdia', b'minf', b'stbl']))
        if not stbl: continue
        sa, sb = stbl[0]
        stsd = list(_find(data, sa, sb, [b'stsd']))[0]
        entry = data[stsd[0] + 8: stsd[1]]
        if entry[4:8] not in (b'avc1', b'avc3'): continue
        # avcC inside the sample entry
        e0 = stsd[0] + 8
        avcc = None
        for typ, a, b in _boxes(data, e0 + 8 + 78, e0 + struct.unpack('>I', entry[:4])[0]):
            if typ == b'avcC': avcc = data[a:b]
        lensize = (avcc[4] & 3) + 1
        n = avcc[5] & 31; p = 6; sps = []; pps = []
        for _ in range(n):
            l = struct.unpack('>H', avcc[p:p+2])[0]; sps.append(avcc[p+2:p+2+l]); p += 2 + l
        m = avcc[p]; p += 1
        for _ in range(m):
            l = struct.unpack('>H', avcc[p:p+2])[0]; pps.append(avcc[p+2:p+2+l]); p += 2 + l
        def full(name):
            a, b = list(_find(data, sa, sb, [name]))[0]
            return data[a:b]
        stsz = full(b'stsz'); ssize, cnt = struct.unpack('>II', stsz[4:12])
        sizes = [ssize] * cnt if ssize else list(struct.unpack('>%dI' % cnt, stsz[12:12 + 4 * cnt]))
        stsc = full(b'stsc'); nsc = struct.unpack('>I', stsc[4:8])[0]
        sc = [struct.unpack('>III', stsc[8 + 12 * i: 20 + 12 * i]) for i in range(nsc)]
        try:
            stco = full(b'stco'); nco = struct.unpack('>I', stco[4:8])[0]
            offs = list(struct.unpack('>%dI' % nco, stco[8:8 + 4 * nco]))
        except IndexError:
            co64 = full(b'co64'); nco = struct.unpack('>I', co64[4:8])[0]
            offs = list(struct.unpack('>%dQ' % nco, co64[8:8 + 8 * nco]))
        samples = []; si = 0
        for ci, co in enumerate(offs):
            per = 0
            for first, spc, _ in sc:
                if ci + 1 >= first: per = spc
            pos = co
            for _ in range(per):
                if si >= len(sizes): break
                samples.append(data[pos:pos + sizes[si]]); pos += sizes[si]; si += 1
        out = []
        for s in samples:
            nals = []; p = 0
            while p + lensize <= len(s):
                l = int.from_bytes(s[p:p + lensize], 'big'); p += lensize
                nals.append(s[p:p + l]); p += l
            out.append(nals)
        return sps, pps, out
    raise ValueError("no avc1 track")
