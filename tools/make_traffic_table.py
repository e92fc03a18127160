"""profiles/<round>_pmc_traffic.json (round = the tag up to its first underscore) from the FETCH_SIZE / WRITE_SIZE summaries of tools/collect_profiles.sh:
   python tools/make_traffic_table.py <tag> [images]      (reads gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE_ex<images>.txt)
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KiB, and FETCH_SIZE counts 64 of every 128 fetched bytes on
gfx950 (tools/ubench/fetch_calib.hip, profiles/r02_a_fetch_calibration.txt). Values are means per dispatch; units_per_launch = images one
dispatch of that kernel covers (FAST goes out in sub-launches over the batch)."""
import sys, os, re, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; images = int(sys.argv[2]) if len(sys.argv) > 2 else 256
def read(counter):
    out = {}; cur = None
    for l in open(os.path.join(ROOT, "gpurun_out", "%s_pmc_%s_ex%d.txt" % (tag, counter, images))):
        if not l.startswith(" "):
            cur = l.strip().replace("viorb::", "")
        else:
            m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\(n=(\d+)\)", l)
            if m and m.group(1) == counter:
                out[cur] = (float(m.group(2)), int(m.group(3)))
    return out
F, W = read("FETCH_SIZE"), read("WRITE_SIZE")
steps = F["k_orient_describe"][1]
names = {"k_resize_stream": "k_resize", "k_resize2": "k_resize", "k_fast_cells3": "k_fast_cells"}
tab = {"source": "profiles/%s_pmc_FETCH_SIZE_ex%d.txt + profiles/%s_pmc_WRITE_SIZE_ex%d.txt (rocprofv3 --pmc, separate passes, tools/extract_times.py %d); HBM bytes = "
                 "2 x FETCH_SIZE + WRITE_SIZE (KiB x 1024): FETCH_SIZE counts 64 of every 128 fetched bytes on gfx950 for 4- and 16-byte-per-lane streams alike "
                 "(profiles/r02_a_fetch_calibration.txt); means per dispatch" % (tag, images, tag, images, images),
       "config": "euroc", "bytes_per_launch": {}, "units_per_launch": {}, "launches_per_step": {}}
for k in sorted(F):
    if not k.startswith("k_") or k not in W:
        continue
    nm = names.get(k, k)
    per_step = F[k][1] / steps
    tab["bytes_per_launch"][nm] = int((2 * F[k][0] + W[k][0]) * 1024)
    tab["launches_per_step"][nm] = round(per_step, 2)
    tab["units_per_launch"][nm] = images if nm in ("k_resize",) else int(round(images / per_step)) if per_step >= 1 else images
rnd = tag.split("_")[0]                      # r04_a -> r04
json.dump(tab, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd), "w"), indent=1)
print(json.dumps(tab, indent=1))
