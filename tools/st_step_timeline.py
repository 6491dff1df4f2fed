"""Per-launch timeline of ONE set-transformer training step from a rocprofv3 --kernel-trace csv (tools/runs/r06d.sh):
    python tools/st_step_timeline.py <kernel_trace.csv> <launches per step> [step index from the end, default 2]
prints start offset, duration and gap to the previous kernel of every launch of that step."""
import csv
import re
import sys


def main(path, per_step, back=2):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))))
    # the bench's steps are the repeating tail: take `per_step` launches ending `back` steps before the end
    end = len(rows) - back * per_step
    step = rows[end - per_step:end]
    t0, prev_end, total = step[0][0], step[0][0], 0
    for s, e, name in step:
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name[:70]}")
        prev_end, total = e, total + (e - s)
    print(f"step: {(step[-1][1] - t0) / 1e3:.1f} us wall, {total / 1e3:.1f} us in kernels, {len(step)} launches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2)
