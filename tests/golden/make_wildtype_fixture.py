"""Extracts the known answers of the reference's wildtype demo into a data fixture.

Reads /root/reference/Demo/Output/wildtype_demo.out (expected output that the
reference's `Demo/demo` script diffs against) and writes wildtype_demo_mxm.json:
the printed entries of input A, input B, C = A*B (GrB_mxm, saxpy3,
Demo/Program/wildtype_demo.c:366) and C<M,struct,replace> = C*C' (GrB_mxm, dot3,
wildtype_demo.c:381), each entry with its position, its 64-character text and
its 4-by-4 block as printed ("%10.1f": one decimal, so the fixture pins every
value to within 0.05), plus the burble line of each GrB_mxm.
Run once in the build container; the JSON is the committed fixture.
"""
import json
import re
import sys

SRC = "/root/reference/Demo/Output/wildtype_demo.out"
HEAD = re.compile(r"^============= printing the WildType matrix: (.+) \((\d+)-by-(\d+) with (\d+) entries\)$")
ENTRY = re.compile(r"^----------- (.+)\((\d+),(\d+)\):$")
SCALAR = re.compile(r"^a wildtype scalar:  \[(.*)\]$")


def main():
    lines = open(SRC).read().splitlines()
    mats = []
    k = 0
    while k < len(lines):
        h = HEAD.match(lines[k])
        if not h:
            k += 1
            continue
        b = next((q for q in range(k, -1, -1) if lines[q].lstrip().startswith("[ GrB_mxm")), None)
        m = {"name": h.group(1), "nrows": int(h.group(2)), "ncols": int(h.group(3)), "entries": []}
        k += 1
        while not lines[k].startswith("============= that was"):
            e = ENTRY.match(lines[k])
            if e:
                s = SCALAR.match(lines[k + 2])
                rows = [[float(v) for v in lines[k + 3 + r].split()] for r in range(4)]
                assert all(len(r) == 4 for r in rows), lines[k]
                m["entries"].append({"i": int(e.group(2)), "j": int(e.group(3)), "text": s.group(1), "stuff": rows})
                k += 7
            else:
                k += 1
        assert len(m["entries"]) == int(h.group(4))
        if m["name"] == "output C":
            m["burble"] = lines[b].strip()[:160]
        mats.append(m)
    names = [m["name"] for m in mats]
    assert names[:4] == ["input A", "input B", "output C", "output C"], names
    mats[2]["name"], mats[3]["name"] = "C = A*B", "C<M,struct,replace> = C*C'"
    json.dump(mats[:4], open(sys.argv[1] if len(sys.argv) > 1 else "wildtype_demo_mxm.json", "w"), indent=1)


if __name__ == "__main__":
    main()
