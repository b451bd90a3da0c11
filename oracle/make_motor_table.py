#!/usr/bin/env python3
"""Regenerate fpyv_amd/data/f80_thrust_table.csv from the reference's bench report.

TEST/BUILD INFRASTRUCTURE - runs only in the build container (needs /root/reference).

The reference keeps the T-Motor F80 bench report in the vendor's raw export format
(/root/reference/config/t_motos_f80_motor_test.csv: '%' suffixes, decimal commas, 10 columns,
merged label cells).  The hot path only consumes two columns of it (throttle %, thrust in grams;
/root/reference/src/utils/components.py:128-136), split into blocks that end at each 100 % row
(/root/reference/src/utils/flight_time_calculator.py:34-39).  This script extracts exactly those
numbers into the build's own three-column schema so the product never needs the raw file.
`fpyv_amd.params.read_motor_test_report` still parses the raw vendor format for user files.
"""
import csv
import os
import sys

SRC = "/root/reference/config/t_motos_f80_motor_test.csv"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpyv_amd", "data",
                   "f80_thrust_table.csv")


def main():
    rows = list(csv.reader(open(SRC, encoding="utf-8")))
    if rows[0][0] == "Type":
        rows = rows[1:]
    block = 0
    out = []
    for r in rows:
        thr = float(r[2].replace("%", ""))
        grams = float(r[3].replace(",", "."))
        out.append((block, thr, grams))
        if thr == 100.0:
            block += 1
    with open(DST, "w", newline="") as f:
        f.write("# T-Motor F80 bench data, normalised: block index, throttle [%], thrust per motor [g]\n")
        f.write("# blocks end at the 100 % row; block 0 = F80 Pro KV1900 / 5055 tri-blade / 24 V\n")
        f.write("block,throttle_pct,thrust_g\n")
        for b, t, g in out:
            f.write(f"{b},{t:g},{g!r}\n")
    print(f"wrote {len(out)} rows, {block} blocks -> {os.path.normpath(DST)}")


if __name__ == "__main__":
    sys.exit(main())
