"""profiles/<tag>_traffic.json from the two PMC passes of scripts/profile_round.sh (pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt).

usage: make_traffic_json.py <dir with the pmc_*.txt> <chunks> <workload> <out.json> <out.txt>
Counters are kilobytes per dispatch.  The guide's gfx950 correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the
bytes of a WIDE COALESCED STREAMING read) is calibrated for that access pattern only, so the table keeps the raw counter
(`fetch_raw_bytes_per_launch`) next to the figure bench.py quotes (`fetch_bytes_per_launch`): doubled for the kernels whose reads are
16-byte-per-lane streams, RAW for the kernels whose reads are scattered narrow accesses (NARROW below: the lookback search's 2-byte
table probes), where doubling is uncalibrated and would overstate the traffic.  Kernel names are mapped to the names bench.py reports
(the library's launch-timer labels)."""
import json
import re
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_sha16  # noqa: E402  (the table is only quoted by bench.py for the kernel sources it was measured on)

d, chunks, workload, out_json, out_txt = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
# kernels whose fetches are dominated by scattered 2- / 4-byte table probes: FETCH_SIZE is quoted raw for them
NARROW = ("enc_lookback_pipe_kernel", "enc_lookback_kernel")   # (the round-5 pipeline reads streams, but the counter reading of its scattered far-candidate reads is kept raw: the smaller, safer figure)
TY = {"unsigned long": "u64", "unsigned int": "u32", "unsigned short": "u16", "unsigned char": "u8"}


def ty(t):
    """number type of a (possibly truncated) template argument: 'unsigned long', 'unsigned lon', ..."""
    t = t.strip()
    for k, v in sorted(TY.items(), key=lambda kv: -len(kv[0])):
        if t == k or (len(t) >= 10 and k.startswith(t)):
            return v
    return TY.get(t, t)


def label(name):
    name = name.strip()
    # the decode walker that publishes its progress and the expanders under it (decode_trail.hip); bench.py reports them as "~dec_walk_kernel<..>"
    # / "~dec_trail_kernel<..>" and strips the "~"
    m = re.match(r"dec_walk_trail_kernel<([^,>]*)", name)
    if m:
        return f"dec_walk_kernel<{ty(m.group(1))}>"
    m = re.match(r"dec_trail_kernel<([^,>]*)(, (t|f))?", name)   # (<L, true>: the expanders of the walker blocks with a two-variable chunk; names are cut at 34 characters)
    if m:
        return ("dec_trail2_kernel" if m.group(3) == "t" else "dec_trail_kernel") + f"<{ty(m.group(1))}>"
    # (pmc_summary.py cuts the names at 34 characters: "dec_walk_kernel<unsigned long, 8u," / "dec_expand_kernel<unsigned long, f")
    m = re.match(r"dec_walk_kernel<([^,>]*), (\d)u", name)
    if m:
        return ("dec_walk_kernel" if m.group(2) == "8" else "dec_walk4_kernel") + f"<{ty(m.group(1))}>"   # (the ordinary first-stage walker adds to the publishing one's label)
    m = re.match(r"dec_expand_kernel<([^,>]*)(, (t|f))?", name)
    if m:
        return ("dec_expand_lb_kernel" if m.group(3) == "t" else "dec_expand_kernel") + f"<{ty(m.group(1))}>"
    m = re.match(r"pco_decode_kernel<([^,>]*)>?", name)
    if m:
        return f"pco_decode_kernel<{ty(m.group(1))}>"
    if name.startswith("enc_walk_kernel"):
        return "enc_walk16_kernel" if name.startswith("enc_walk_kernel<16") else "enc_walk_kernel"
    if name.startswith("enc_lookback_pipe_kernel"):   # LbPipe<kSmall, kProps, kFastD[, pages]>: the launch-timer labels of pco_gfx_encode_api.inc
        m = re.match(r"enc_lookback_pipe_kernel<LbPipe<(t|f)\w*(, (t|f)\w*)?(, (t|f)\w*)?", name)
        small = bool(m) and m.group(1) == "t"; props = bool(m) and m.group(3) == "t"; fastd = bool(m) and m.group(5) == "t"
        if props:
            return "enc_lookback_pipe_kernel<" + ("small," if small else "") + "props" + (",fastd" if fastd else "") + ">"
        return "enc_lookback_pipe_kernel<small>" if small else "enc_lookback_pipe_kernel"
    if name.startswith("enc_lookback_kernel"):
        return "enc_lookback_kernel<small>" if name.startswith("enc_lookback_kernel<LbCfg<256") else "enc_lookback_kernel"
    if name.startswith("enc_split_kernel"):
        return {"<true, false>": "enc_split_kernel<c16>", "<false, true>": "enc_split_kernel(redo)"}.get(name[len("enc_split_kernel"):], "enc_split_kernel")
    if name.startswith("enc_hist_select_kernel"):
        return "enc_hist_select_kernel"
    m = re.match(r"enc_pack1_kernel<([^,>]*), (\d)u, (t|f)", name)   # <L, mask, wide>: the launch-timer labels of pco_gfx_encode_api.inc (PCO_PACK1)
    if m:
        tags = ([{"6": "sec", "3": "lb"}[m.group(2)]] if m.group(2) in "63" else []) + (["wide"] if m.group(3) == "t" else [])
        return "enc_pack1_kernel" + ("<" + ",".join(tags) + ">" if tags else "")
    m = re.match(r"enc_hist_wide_kernel<(\d+)u>", name)
    if m:
        return f"enc_hist_wide_kernel<{m.group(1)}>"
    return name


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.e+]+)\s+\(", line)
        if m and not line.startswith("kernel"):
            k = label(m.group(1)); old = rows.get(k, (0.0, 0.0))   # (instantiations that share a launch label add up)
            rows[k] = (old[0] + float(m.group(4)), old[1] + float(m.group(3)))
    return rows


f, w = parse(f"{d}/pmc_FETCH_SIZE.txt"), parse(f"{d}/pmc_WRITE_SIZE.txt")
kern = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1])):
    raw = int(f.get(k, (0, 0))[0] * 1000); streaming = not k.startswith(NARROW)
    kern[k] = {"fetch_raw_bytes_per_launch": raw, "fetch_bytes_per_launch": 2 * raw if streaming else raw, "fetch_doubled": streaming,
               "write_bytes_per_launch": int(w.get(k, (0, 0))[0] * 1000)}
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python bench.py --steps 2 --warmup 1 --chunks {chunks} "
                     "--no-cpu-baseline; counters are kilobytes per dispatch; FETCH_SIZE doubled (gfx950 note, MI355X_MICROARCH.md HBM section) for the "
                     "streaming kernels only (fetch_doubled), raw for scattered narrow reads; fetch_raw_bytes_per_launch is the counter as reported",
           "chunks": chunks, "workload": workload, "csrc_sha16": csrc_sha16(), "kernels": kern}, open(out_json, "w"), indent=1)
with open(out_txt, "w") as o:
    o.write(open(f"{d}/pmc_FETCH_SIZE.txt").read()); o.write(open(f"{d}/pmc_WRITE_SIZE.txt").read())
    o.write("\nper launch, bytes (FETCH_SIZE x 1000 raw; x 2 where the reads are wide coalesced streams; WRITE_SIZE x 1000):\n")
    for k, v in kern.items():
        o.write(f"{k.ljust(34)} fetch raw {v['fetch_raw_bytes_per_launch'] / 1e9:8.3f} GB  quoted {v['fetch_bytes_per_launch'] / 1e9:8.3f} GB ({'x2' if v['fetch_doubled'] else 'raw'})  write {v['write_bytes_per_launch'] / 1e9:8.3f} GB\n")
