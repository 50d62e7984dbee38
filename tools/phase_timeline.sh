#!/bin/bash
# tools/phase_timeline.sh <outdir> [probe args] -- the two-group schedule's timeline (opts.phases = 2) from a rocprofv3
# kernel trace: for a window of rounds in the middle of the run, start / end of every k_round_mc and k_ph_mark launch
# relative to the window's start, and the averages: launch duration, period, the gap between a group's mark step and its
# next round kernel, how long both round kernels run side by side.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
: "${1:?usage: see the header comment}"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; shift; mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$O/prof" -o p -- python tools/scale_probe.py ${@:-100000000,150,65536} > "$O/run.log" 2>&1 || true
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    kind = ("R" if "k_round" in k else "M" if ("k_ph_mark" in k or "k_mg_mark" in k) else "T" if "k_trim_bins" in k else
            "L" if "k_long" in k else "A" if "alt_resolve" in k else None)
    if kind:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?")))
ev.sort()
with open(O + "/phase_timeline.txt", "w") as o:
    R = [e for e in ev if e[2] == "R"]
    M = [e for e in ev if e[2] == "M"]
    o.write("round kernels %d (avg %.1f us), mark steps %d (avg %.1f us), span %.1f ms\n" % (
        len(R), sum(e[1] - e[0] for e in R) / max(len(R), 1) / 1e3, len(M), sum(e[1] - e[0] for e in M) / max(len(M), 1) / 1e3,
        (ev[-1][1] - ev[0][0]) / 1e6))
    # busy time: union of round-kernel intervals
    busy, cur_s, cur_e, both = 0, None, None, 0
    for s, e, _, _ in R:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            both += min(e, cur_e) - s
            cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    o.write("union of round-kernel time %.1f ms, two running side by side %.1f ms\n" % (busy / 1e6, both / 1e6))
    mid = len(ev) // 2
    t0 = ev[mid][0]
    o.write("window from the middle of the run (us from its start; queue):\n")
    for s, e, k, q in ev[mid:mid + int(__import__('os').environ.get('TL_ROWS', '48'))]:
        o.write("  %s q%-3s %9.1f .. %9.1f  (%6.1f)\n" % (k, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
rm -rf "$O/prof"
cat "$O/phase_timeline.txt"
