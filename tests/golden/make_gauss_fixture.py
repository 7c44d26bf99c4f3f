"""Extracts the known answers of the reference's gauss demo into a data fixture.

Reads /root/reference/Demo/Output/gauss_demo.out (expected output that the
reference's `Demo/demo` script diffs against) and writes gauss_demo_mxm.json:
for each of the seven GrB_mxm calls of Demo/Program/gauss_demo.c:402-470 the
method named in the burble line and the printed 4x4 result (null = no entry).
Run once in the build container; the JSON is the committed fixture.
"""
import json
import re
import sys

SRC = "/root/reference/Demo/Output/gauss_demo.out"
HEADERS = [
    ("A = A^2", "=============== Gauss A = A^2 matrix:"),
    ("C = diag(AA')", "=============== Gauss C = diag(AA') matrix:"),
    ("C = D*A", "=============== Gauss C = D*A matrix:"),
    ("C = A*D", "=============== Gauss C = A*D matrix:"),
    ("C += A*A'", "=============== Gauss C += A*A' matrix:"),
    ("C += B*A", "=============== Gauss C += B*A:"),
    ("C += A*B", "=============== Gauss C += A*B:"),
]
ENTRY = re.compile(r"\(\s*(-?\d+),\s*(-?\d+)\)|\s{6}\.\s{5}")


def main():
    lines = open(SRC).read().splitlines()
    out = []
    for name, header in HEADERS:
        at = next(k for k, l in enumerate(lines) if l.strip() == header)
        # the burble line of the producing GrB_mxm is the closest "[ GrB_mxm" above the header
        b = next(k for k in range(at, -1, -1) if lines[k].lstrip().startswith("[ GrB_mxm"))
        rows = []
        k = at
        while len(rows) < 4:
            k += 1
            if lines[k].startswith("row "):
                body = lines[k].split(":", 1)[1]
                row = []
                for m in ENTRY.finditer(body):
                    row.append(None if m.group(1) is None else [int(m.group(1)), int(m.group(2))])
                assert len(row) == 4, lines[k]
                rows.append(row)
        out.append({"name": name, "burble": lines[b].strip(), "matrix": rows})
    json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gauss_demo_mxm.json", "w"), indent=1)


if __name__ == "__main__":
    main()
