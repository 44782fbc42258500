"""figures.jsonl (tests/test_full_size.py under ADM_ACCURACY_LOG, run with the F(4x4) kernel on and off) -> the markdown table of profiles/r06_accuracy.md"""
import json
import sys

rows = {}
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    rows.setdefault(r["what"], {})[r["wino6"]] = r


def cell(r):
    if r is None:
        return "—"
    if "value" in r:
        return f"{r['value']:.2e} ({r['value'] / r['bar'] * 100:.1f} % of the bar)"
    s = f"{r['float_err']:.2e} ({r['float_err'] / r['bar'] * 100:.1f} % of the bar); {r['max_lsb']} LSB max, {r['identical_pixels'] * 100:.3f} % identical"
    if "perturbation_growth" in r:
        s += f"; 1e-6 start perturbation -> {r['perturbation_growth']:.1e}"
    return s


print("| figure (bar) | F(4x4) kernel on (default layer rule) | off (`ADM_WINO6=0`: F(2x2,3x3) everywhere) |")
print("|---|---|---|")
for what, by in rows.items():
    any_r = next(iter(by.values()))
    print(f"| {what} (bar {any_r['bar']:.0e}) | {cell(by.get('default'))} | {cell(by.get('0'))} |")
