"""A window of consecutive kernel launches from a rocprofv3 --kernel-trace csv: start offset, duration, gap to the previous launch.
    python tools/kernel_window.py <kernel_trace.csv> <first launch (negative: from the end)> <count>"""
import csv
import re
import sys


def main(path, first, count):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))))
    if first < 0:
        first += len(rows)
    win = rows[first:first + count]
    t0, prev, busy = win[0][0], win[0][0], 0
    for s, e, name in win:
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {name[:80]}")
        prev, busy = e, busy + (e - s)
    print(f"window: {(win[-1][1] - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us in kernels, {len(win)} launches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
