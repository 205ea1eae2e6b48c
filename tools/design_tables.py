#!/usr/bin/env python3
"""Regenerate the numeric tables of DESIGN.md from the bench lines committed under profiles/ (figures
that live in profiles/ are generated into the document, not copied by hand).

  python tools/design_tables.py            rewrite the block between the GENERATED markers of DESIGN.md
  python tools/design_tables.py --check    exit 1 if DESIGN.md is not what the profiles give (tests/test_host_logic.py runs this)
  python tools/design_tables.py --print    the block on stdout
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
BEGIN, END = "<!-- BEGIN GENERATED: tools/design_tables.py -->", "<!-- END GENERATED -->"
BEGIN2, END2 = "<!-- BEGIN GENERATED frames-in-flight: tools/design_tables.py -->", "<!-- END GENERATED frames-in-flight -->"


def flight_block():
    """The frames-in-flight table of section 5.2 from profiles/r03_frames_in_flight.txt (tools/pipeline_bench.sh)."""
    rows = {}
    with open(os.path.join(PROF, "r03_frames_in_flight.txt")) as f:
        for line in f:
            m = re.match(r"(--size .*?--frames-in-flight (\d))\s+([\d.]+) Mrays/s\s+([\d.]+) ms/frame.*region ([\d.]+|None)", line)
            if m:
                key = re.sub(r"\s*--frames-in-flight \d", "", m.group(1)).strip()
                rows.setdefault(key, {})[int(m.group(2))] = (float(m.group(3)), float(m.group(4)), m.group(5))
    names = {"--size 4096": ("4096² whole frame", "rate"), "--size 2048": ("2048² whole frame (config 3)", "rate"),
             "--size 4096 --emulate-parts 2 --emulate-part 1": ("one rank of 2 (4096²)", "ms"),
             "--size 4096 --emulate-parts 4 --emulate-part 1": ("one rank of 4", "ms"),
             "--size 4096 --emulate-parts 8 --emulate-part 1": ("one rank of 8", "ms"),
             "--size 4096 --integrator dp45 --precision 64": ("4096² DP45 float64", "rate")}
    out = [BEGIN2, "", "| workload | F = 1 | F = 2 | F = 3 | chip-level issue fraction of the region, F = 1 → best |", "|---|---|---|---|---|"]
    for key, (label, kind) in names.items():
        r = rows.get(key)
        if not r:
            continue
        def cell(F):
            if F not in r:
                return "—"
            v, ms, _ = r[F]
            return f"{v:.0f} Mrays/s ({ms:.2f} ms)" if kind == "rate" else f"{ms:.2f} ms"
        regs = [float(r[F][2]) for F in sorted(r) if r[F][2] != "None"]
        out.append(f"| {label} | {cell(1)} | {cell(2)} | {cell(3)} | " + (f"{regs[0]:.2f} → {max(regs):.2f}" if regs else "—") + " |")
    r8 = rows.get("--size 4096 --emulate-parts 8 --emulate-part 1", {})
    if 1 in r8 and 3 in r8:
        out += ["", f"(one rank of 8 with three frames in flight renders its share of a frame every {r8[3][1]:.2f} ms: 8 ranks ≈ "
                    f"{4096 * 4096 / r8[3][1] / 1e3:.0f} Mrays/s before the gather, against ≈ {4096 * 4096 / r8[1][1] / 1e3:.0f} one frame at a time)"]
    out += ["", END2]
    return "\n".join(out)


def load(name):
    with open(os.path.join(PROF, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def row_of(tag, d):
    r, c = d["roofline"], d["config"]
    ex = r.get("executed", {})
    frac = f"{r['frac']:.3f} / {r['frac_at_held_clock']:.3f} @ {ex.get('clock_mhz_held', 0):.0f} MHz" if r.get("frac") else "n/a (no PMC count)"
    traffic = f"{r['traffic'] / 1e6:.0f} / {r['algorithmic_bytes_per_launch'] / 1e6:.0f}" if r.get("traffic") else f"– / {r['algorithmic_bytes_per_launch'] / 1e6:.0f}"
    pipe = f"{d['pipelined']['value']:.0f}" if d.get("pipelined") and "value" in d["pipelined"] else "–"
    return (f"| `{tag}` | {c['workload']} ({d['dtype']}, {c['schedule']}) | **{d['value']:.0f}** | {d['ms_per_step']:.3f} | {r['avg_launch_ms']:.3f} | "
            f"{r['other_kernels_ms']['prologue']:.3f} / {r['other_kernels_ms']['epilogue']:.3f} | {frac} | {traffic} | {pipe} |")


def block():
    R = "r03"
    out = [BEGIN, "",
           f"Every figure of this block is read from `profiles/{R}_bench_*.json` (bench lines measured on one MI355X by "
           "`tools/refresh_profiles.sh measure`); `python tools/design_tables.py --check` fails when the document and the profiles disagree.", ""]
    files = [("default", "the driver's line"), ("2048", "config 3"), ("imagelens", "config 4"), ("queue", "queue schedule"), ("dp45", "DP45, float32 controller")]
    out += ["| profile | workload | Mrays/s | ms / frame | K2 ms / launch | K1 / K3 ms | issue frac nominal / at held clock | HBM MB per launch measured / algorithmic | 3 frames in flight, Mrays/s |",
            "|---|---|---|---|---|---|---|---|---|"]
    lines = {}
    for tag, _ in files:
        try:
            lines[tag] = load(f"{R}_bench_{tag}.json")
        except OSError:
            continue
        out.append(row_of(f"{R}_bench_{tag}.json", lines[tag]))
    d = lines["default"]
    p = d.get("production_path")
    if p:
        pr = p["roofline"]
        out += ["", f"**Production path** (`production_path` of `{R}_bench_default.json`: the same 4096² frame, DP45 float64 with the reference's float64 step "
                    f"controller, {p['steps']} frames after {p['warmup']} warm-up): **{p['value']:.0f} Mrays/s**, {p['ms_per_step']:.3f} ms per frame; "
                    f"`k_kerr_direct<double, Dp45<double, true>>` {pr['avg_launch_ms']:.3f} ms per launch, {pr['executed']['valu_per_wave_iter']:.0f} VALU wave-instructions per "
                    f"wave iteration x {pr['executed']['wave_iters_per_launch']:,} iterations = {pr['executed']['valu_wave_insts_per_launch'] / 1e9:.2f} G per launch against "
                    f"1024 SIMDs x 2.4 GHz / 4 cycles: **issue frac {pr['frac']:.3f}** ({pr['frac_at_held_clock']:.3f} at the {pr['executed']['clock_mhz_held']:.0f} MHz held); "
                    f"HBM {pr['traffic'] / 1e6:.0f} MB per launch measured against {pr['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic; "
                    f"{p['config']['mean_dp45_attempts_per_ray']} step attempts and {p['config']['mean_rhs_evals_per_ray']} right-hand sides per ray; "
                    f"K1 / K3 {pr['other_kernels_ms']['prologue']:.3f} / {pr['other_kernels_ms']['epilogue']:.3f} ms."]
    cb, pcb = d.get("cpu_baseline"), (p or {}).get("cpu_baseline")
    if cb:
        out += ["", f"**CPU baseline in the same run** (oracle, perf build, {cb['cores']} OpenMP threads = the job's quota of {cb['host_cpus_visible']} visible CPUs): "
                    f"RK4 float64 **{cb['value']:.3f} Mrays/s** ({cb['sample'].split(',')[0]})"
                    + (f"; DP45 float64 **{pcb['value']:.3f} Mrays/s** ({pcb['mean_rhs_evals_per_ray']:.0f} right-hand sides per ray against {cb['mean_rhs_evals_per_ray']:.0f})." if pcb else ".")]
    if d.get("chain_floor"):
        c = d["chain_floor"]
        out += ["", f"**Serial chain**: the frame's longest ray ({c['longest_ray_steps']} steps, pixel {tuple(c['pixel'])}) traced alone on the chip: {c['alone_ms']:.2f} ms = "
                    f"{c['us_per_step']:.3f} µs per step."]
    if d.get("end_to_end_ms"):
        e = d["end_to_end_ms"]
        out += [f"**Host-pointer frame** (`lt_render`, RGBA8 out, PCIe included): {e['pinned_dst_ms']:.2f} ms into pinned memory, {e['pageable_dst_ms']:.2f} ms into pageable."]
    pj = d.get("projected_ranks")
    if pj and "error" not in pj:
        out += ["", "**Projected ranks** (`projected_ranks`: rank p of n renders its block-cyclic rows alone on this one GPU, back to back; a PROJECTION of the compute of an "
                    "n-GPU frame, no gather, not a multi-GPU measurement):", "",
                "| integrator | slowest rank of 2, ms (Mrays/s) | of 4 | of 8 | every rank of 8, ms |", "|---|---|---|---|---|"]
        for integ, v in pj.items():
            cells = [f"{v[n]['slowest_rank_ms']:.2f} ({v[n]['mrays_per_s_before_gather']:.0f})" for n in ("2", "4", "8")]
            out.append(f"| {integ} | " + " | ".join(cells) + " | " + " ".join(f"{x:.2f}" for x in v["8"]["frame_ms_per_rank"]) + " |")
    out += ["", END]
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    with open(path) as f:
        doc = f.read()
    new = block()
    if "--print" in sys.argv:
        print(new)
        print(flight_block())
        return 0
    stale = False
    for begin, end, text in ((BEGIN, END, new), (BEGIN2, END2, flight_block())):
        m = re.search(re.escape(begin) + r".*?" + re.escape(end), doc, re.S)
        if not m:
            print(f"DESIGN.md has no block {begin}", file=sys.stderr)
            return 1
        stale = stale or m.group(0) != text
        doc = doc[:m.start()] + text + doc[m.end():]
    if "--check" in sys.argv:
        if stale:
            print("DESIGN.md's generated blocks are stale: run python tools/design_tables.py", file=sys.stderr)
            return 1
        return 0
    with open(path, "w") as f:
        f.write(doc)
    return 0


if __name__ == "__main__":
    sys.exit(main())
