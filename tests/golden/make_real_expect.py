#!/usr/bin/env python3
"""Expectations for the two third-party clips the container image holds
(tests/test_cabac_real_streams.py): per slice the macroblock count, the
position the arithmetic decoder stopped at and the checksum over the full
macroblock syntax, from the oracle's CABAC decoder.  The clips themselves are
not ours and are not committed: they are identified by their SHA-256.

    python tests/golden/make_real_expect.py  ->  tests/golden/real_streams_expect.json
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import orc  # noqa: E402
import test_cabac_real_streams as t  # noqa: E402


def main():
    out = {}
    for name, q in (("realshort.mp4", False), ("cockatoo.mp4", True)):
        path = t.sample(name)
        head, nals = t.slices_of(path)
        try:
            t.quirk(q)
            ref = orc.parse_nalus(head + nals, flags=t.FLAGS, want=0)
        finally:
            t.quirk(False)
        out[name] = {
            "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
            "x264_before_151": q,
            "slices": [[r.status, r.mb_count, r.end_bit, "%016x" % r.checksum]
                       for r in ref.results]}
    with open(os.path.join(HERE, "real_streams_expect.json"), "w") as f:
        f.write("{\n")
        names = list(out)
        for i, n in enumerate(names):
            e = out[n]
            f.write(' "%s": {"sha256": "%s", "x264_before_151": %s, "slices": [\n'
                    % (n, e["sha256"], json.dumps(e["x264_before_151"])))
            for j, r in enumerate(e["slices"]):
                f.write('  [%d, %d, %d, "%s"]%s\n' %
                        (r[0], r[1], r[2], r[3],
                         "," if j + 1 < len(e["slices"]) else ""))
            f.write(" ]}%s\n" % ("," if i + 1 < len(names) else ""))
        f.write("}\n")


if __name__ == "__main__":
    main()
