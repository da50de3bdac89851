"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into profiles/r01_conv_hbm_traffic.json.

usage: python tools/pmc_aggregate.py <fetch_dir> <write_dir> <out.json>
Counters are KiB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE reports half of a 16 B/lane coalesced read
stream, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is used as is."""
import csv, glob, json, os, re, sys
from collections import defaultdict

# every forward / data-gradient launch with a 128-wide output tile (what bench.py's roofline object times)
DOMINANT = re.compile(r"conv_fwd_dma_kernel<(128|64), 128, 2, (true|false), [01](, 256)?>|conv_fwd_dma32_kernel<128, true, 0, 3")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def load(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                k = short(r["Kernel_Name"])
                per[k][0] += 1
                per[k][1] += float(r["Counter_Value"])
    return per


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    dom = sorted(k for k in fetch if DOMINANT.search(k))
    n = sum(fetch[k][0] for k in dom)
    f_kib = sum(fetch[k][1] for k in dom)
    w_kib = sum(write[k][1] for k in dom if k in write)
    top = lambda per: {k: {"launches": v[0], "avg_mib_per_launch": v[1] / 1024 / max(v[0], 1)}
                       for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]}
    res = {
        "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline ; same with --pmc WRITE_SIZE (two separate passes); tools/pmc_aggregate.py",
        "correction": "MI355X_MICROARCH.md HBM section: counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of a 16 B/lane coalesced read stream (buffer_load ... lds here) -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is",
        "dominant_kernels": dom, "launches": n,
        "fetch_bytes_per_launch_raw": f_kib * 1024 / max(n, 1),
        "read_bytes_per_launch_corrected": 2 * f_kib * 1024 / max(n, 1),
        "write_bytes_per_launch": w_kib * 1024 / max(n, 1),
        "hbm_bytes_per_launch": (2 * f_kib + w_kib) * 1024 / max(n, 1),
        "per_kernel": {"FETCH_SIZE": top(fetch), "WRITE_SIZE": top(write)},
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("launches", "read_bytes_per_launch_corrected", "write_bytes_per_launch", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
