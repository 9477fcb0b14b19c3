"""Diagnostic (tooling): busy time and idle gaps of the last steps in a rocprofv3 kernel trace of tools/diag/ref_loop_trace.py.
usage: python tools/diag/ref_loop_gaps.py DIR [n_steps]"""
import csv, glob, sys, collections
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps: split at k_patch_fwd occurrences (one per step); keep the last n
idx = [i for i, e in enumerate(ev) if "k_patch_fwd" in e[2]]
idx = idx[-(n + 1):]
ev = ev[idx[0]:idx[-1]]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
gaps = collections.Counter(); gapn = collections.Counter()
prev_end, prev_name = ev[0][1], ev[0][2]
for s, e, k in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = s - cur_e
        key = (prev_name[:40], k[:40])
        gaps[key] += g; gapn[key] += 1
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= prev_end:
        prev_end, prev_name = e, k
busy += cur_e - cur_s
steps = len(idx) - 1
print(f"{steps} steps: span {span / steps / 1e3:.1f} us/step, busy (union of kernels) {busy / steps / 1e3:.1f} us/step, idle {(span - busy) / steps / 1e3:.1f} us/step")
print(f"kernels/step {len(ev) / steps:.1f}; sum of durations {sum(e - s for s, e, _ in ev) / steps / 1e3:.1f} us/step")
print("largest idle gaps (us/step, count/step, before-kernel <- after-kernel):")
for key, g in gaps.most_common(14):
    print(f"  {g / steps / 1e3:7.1f} {gapn[key] / steps:5.1f}   {key[0]}  ->  {key[1]}")
dur = collections.Counter(); cnt = collections.Counter()
for s, e, k in ev:
    dur[k[:60]] += e - s; cnt[k[:60]] += 1
print("top kernels (us/step, launches/step):")
for k, v in dur.most_common(16):
    print(f"  {v / steps / 1e3:7.1f} {cnt[k] / steps:5.1f}  {k}")
