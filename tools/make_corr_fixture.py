"""Extracts the numbers of the reference's committed data/output/official/correlations.xlsx (calccorr's output,
src/Hmc.jl:1094-1163: per end date the correlation matrix of the per-draw columns mu | sigma | pi | vec(A) | forecast_12 over
the 250 000 kept draws of the production run) into two CSV fixtures:
  tests/golden/official_correlations_forecast_row.csv   Sheet1: per date the forecast's correlation with every column (455 x 19)
  tests/golden/official_correlations_matrices.csv       the full 19 x 19 matrix of every 12th date (date, row label, 19 values)
The workbook is read as a zip of XML (zipfile + ElementTree: nothing in it is executed).  DATA derived from a reference output
-- runs only where /root/reference exists; the GPU box uses the committed CSVs.   python tools/make_corr_fixture.py [/root/reference]"""
import csv
import os
import sys
import zipfile
import xml.etree.ElementTree as ET

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
M = "{http://schemas.openxmlformats.org/spreadsheetml/2006/main}"
R = "{http://schemas.openxmlformats.org/officeDocument/2006/relationships}"
z = zipfile.ZipFile(os.path.join(ref, "data", "output", "official", "correlations.xlsx"))
strings = ["".join(t.text or "" for t in si.iter(M + "t")) for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall(M + "si")]
rid = {r.get("Id"): r.get("Target") for r in ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))}
sheets = [(s.get("name"), "xl/" + rid[s.get(R + "id")]) for s in ET.fromstring(z.read("xl/workbook.xml")).find(M + "sheets")]


def cells(path):
    """rows of a sheet as lists of python values (shared strings resolved, numbers as their original text)"""
    out = []
    for row in ET.fromstring(z.read(path)).find(M + "sheetData"):
        vals = {}
        for c in row:
            v = c.find(M + "v")
            col = "".join(ch for ch in c.get("r") if ch.isalpha())
            idx = 0
            for ch in col:
                idx = idx * 26 + ord(ch) - 64
            vals[idx] = None if v is None else (strings[int(v.text)] if c.get("t") == "s" else v.text)
        out.append([vals.get(i) for i in range(1, max(vals) + 1)] if vals else [])
    return out


ASCII = {"μ": "mu", "σ": "sigma", "π": "pi"}
asc = lambda s: "".join(ASCII.get(ch, ch) for ch in s)
first = cells(sheets[0][1])
labels = [asc(x) for x in first[0][1:]]
with open(os.path.join(GOLDEN, "official_correlations_forecast_row.csv"), "w", newline="") as fh:
    w = csv.writer(fh, lineterminator="\n")
    w.writerow(["date"] + labels)
    for r in first[1:]:
        w.writerow(r)
n = 0
with open(os.path.join(GOLDEN, "official_correlations_matrices.csv"), "w", newline="") as fh:
    w = csv.writer(fh, lineterminator="\n")
    w.writerow(["date", "row"] + labels)
    for i, (name, path) in enumerate(sheets[1:]):
        if i % 12:
            continue
        rows = cells(path)
        assert [asc(x) for x in rows[0][1:]] == labels and len(rows) == 20
        date = name.replace("_", "-") + "-01"
        for r in rows[1:]:
            w.writerow([date, asc(r[0])] + r[1:])
        n += 1
print("Sheet1: %d dates; %d full matrices; labels %s" % (len(first) - 1, n, labels))
