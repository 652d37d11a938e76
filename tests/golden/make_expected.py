#!/usr/bin/env python3
"""Extracts the known-answer values the reference's RUnit tests hold for the
hot path into tests/golden/expected.json (data only; no reference code).

Run in the build container (needs /root/reference):
    python tests/golden/make_expected.py
Sources (reference inst/unitTests/):
    test_generateCytosineReport.R:1-260, test_generateMhlReport.R:1-123,
    test_generateCytosineReport.R:262-433 (long-read MM/ML cases: inputs and expected tables),
    test_extractPatterns.R:1-270 (every numeric expectation, in file order),
    test_simulateBam.R:53-87, test_generateBedReport.R:12-83,
    test_preprocessBam.R:11-15
plus the survey-time probe outputs recorded in SURVEY.md section 8c.
"""
import json
import os
import re

REF = "/root/reference/inst/unitTests"
HERE = os.path.dirname(os.path.abspath(__file__))


def blocks(fn):
    """Yield (expr, value) source strings of every RUnit::checkEquals(expr, value) call."""
    src = open(os.path.join(REF, fn)).read()
    for m in re.finditer(r"RUnit::checkEquals\(", src):
        i = m.end()
        depth, args, cur = 1, [], []
        while depth > 0:
            ch = src[i]
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append("".join(cur))
                cur = []
            else:
                cur.append(ch)
            i += 1
        args.append("".join(cur))
        yield [re.sub(r"\s+", " ", a).strip() for a in args]


def nums(s):
    s = s.strip()
    if s.startswith("c("):
        s = s[2:-1]
    return [float(x) if ("." in x or "e" in x.lower()) else int(x) for x in re.findall(r"-?\d+\.?\d*(?:[eE]-?\d+)?", s)]


def collect(fn):
    out = []
    for a in blocks(fn):
        if len(a) >= 2 and re.match(r"^(c\()?[-\d\s.,eE]+\)?$", a[1]):
            out.append({"expr": a[0], "value": nums(a[1])})
    return out


def collect_na(fn):
    """As collect(), but integer vectors may hold NA (-> null) and a:b ranges (test_extractPatterns.R)."""
    out = []
    for a in blocks(fn):
        if len(a) < 2 or not re.match(r"^(c\()?[-\d\s.,:NA]+\)*$", a[1]):
            continue
        vals = []
        for tok in re.findall(r"NA|-?\d+:-?\d+|-?\d+", a[1]):
            if tok == "NA":
                vals.append(None)
            elif ":" in tok:
                lo, hi = tok.split(":")
                vals += list(range(int(lo), int(hi) + 1))
            else:
                vals.append(int(tok))
        out.append({"expr": a[0], "value": vals})
    return out



def long_read_cases():
    """test_generateCytosineReport.R:262-433: simulateBam() inputs (flag, seq, Mm, Ml), the generateCytosineReport
    options and every expected value of the long-read (MM/ML) section, in file order."""
    src = open(os.path.join(REF, "test_generateCytosineReport.R")).read()
    sec = src[src.index("### long-read"):]
    sec = sec[:sec.index("  # simulateBam(")]

    def call_args(text, start):
        """text[start] is just after '('; returns (argument string, index after the closing paren)."""
        depth, i = 1, start
        while depth:
            depth += text[i] in "([" 
            depth -= text[i] in ")]"
            i += 1
        return text[start:i - 1], i

    def split_args(a):
        out, cur, depth, q = [], [], 0, False
        for ch in a:
            if ch == '"':
                q = not q
            if not q:
                depth += ch in "(["
                depth -= ch in ")]"
                if ch == "," and depth == 0:
                    out.append("".join(cur).strip())
                    cur = []
                    continue
            cur.append(ch)
        out.append("".join(cur).strip())
        return out

    def ints(t):
        m = re.match(r"^(?:as\.integer\()?rep\.int\((-?\d+), *(\d+)\)\)?$", t.strip())
        if m:
            return [int(m.group(1))] * int(m.group(2))
        return [int(x) for x in re.findall(r"-?\d+", t)]

    def strs(t):
        return re.findall(r'"([^"]*)"', t)

    def table(t):
        inner, _ = call_args(t, t.index("(") + 1)
        cols = {}
        for a in split_args(inner):
            k, v = a.split("=", 1)
            k = k.strip()
            if k == "strand":
                cols[k] = strs(split_args(v[v.index("(") + 1:])[0])
            elif k == "context":
                cols[k] = ints(split_args(v[v.index("(") + 1:])[0])
            else:
                cols[k] = ints(v)
        return cols

    cases, cur, opts = [], None, None
    for m in re.finditer(r"simulateBam\(|generateCytosineReport\(|RUnit::checkEquals\(", sec):
        a, _ = call_args(sec, m.end())
        a = re.sub(r"\s+", " ", a)
        if m.group(0).startswith("simulateBam"):
            kv = dict(x.split("=", 1) for x in split_args(a))
            kv = {k.strip(): v.strip() for k, v in kv.items()}
            ml = [ints(x) for x in split_args(call_args(kv["Ml"], kv["Ml"].index("(") + 1)[0])]
            cur = {"flag": ints(kv["flag"]), "seq": strs(kv["seq"]), "pos": ints(kv["pos"]), "Mm": strs(kv["Mm"]), "Ml": ml,
                   "reports": []}
            cases.append(cur)
        elif m.group(0).startswith("generateCytosineReport"):
            kv = dict(x.split("=", 1) for x in split_args(a)[1:])
            opts = {"min_prob": int(kv.get("min.prob", "-1")), "highest_prob": kv.get("highest.prob", "TRUE").strip() == "TRUE",
                    "report_context": strs(kv["report.context"])[0], "checks": []}
            cur["reports"].append(opts)
        else:
            expr, val = split_args(a)[:2]
            opts["checks"].append({"expr": expr, "value": table(val) if val.startswith("data.table") else ints(val)})
    return cases


exp = {
    "_source": "reference inst/unitTests/*.R (RUnit known-answer values), extracted by tests/golden/make_expected.py",
    "generateCytosineReport": collect("test_generateCytosineReport.R"),
    "generateMhlReport": collect("test_generateMhlReport.R"),
    "simulateBam": collect("test_simulateBam.R"),
    "generateBedReport": collect("test_generateBedReport.R"),
    "longRead": long_read_cases(),
    "extractPatterns": collect_na("test_extractPatterns.R"),
    # Recorded at survey time from the unmodified reference objects;
    # not reproducible in this image (reference unbuildable), kept as extra pins.
    "survey_probe": {
        "amplicon010meth": {"templates": 500, "bytes": 190455, "pass": 48, "cg_thr": [478, 632, 6449],
                            "cg_thr_strand_rows": [241, 237], "cg_nothr": [478, 683, 6398],
                            "cx_nothr": [5936, 893, 43660], "cx_ctx_rows": [3949, 1509, 478]},
        "amplicon100meth": {"pass": 473, "cg_thr": [454, 6325, 439]},
        "amplicon000meth": {"pass": 5, "cg_thr": [543, 15, 6971]},
        "dragen-pe-namesort-xg-xm": {"templates": 100, "cx_nothr": [3827, 435, 5214], "cx_ctx_rows": [2608, 920, 299]},
        "capture": {"templates": 2968, "pass": 1020},
    },
}
with open(os.path.join(HERE, "expected.json"), "w") as f:
    json.dump(exp, f, indent=1)
for k in ("generateCytosineReport", "generateMhlReport", "simulateBam", "generateBedReport"):
    print(k, len(exp[k]))
