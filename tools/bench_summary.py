#!/usr/bin/env python3
"""One screen of the numbers DESIGN.md's claim table quotes, read from the committed lines of a round: python tools/bench_summary.py r06"""
import glob
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for f in sorted(glob.glob(os.path.join(P, f"{R}_bench_n1_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    b = r.get("beyond_mall") or {}; lf = r.get("launch_time_fit") or {}; s = r.get("sustained") or {}
    print(os.path.basename(f)[len(R) + 10:-5], f"value {d['value'] / 1e9:.2f} G  wall {d['ms_per_step'] * 1e3:.2f} us/step  events {r['avg_launch_us']:.2f} us/launch  "
          f"frac {r['frac']:.3f}  frac_events {r.get('frac_events', 0):.3f}  bound {r['bound']}  host {r.get('host_enqueue_us', 0):.1f} us")
    if s:
        print(f"    sustained {s['avg_launch_us']:.2f} us {s['frac']:.3f}")
    if b:
        print(f"    2^23: {b['avg_launch_us']:.1f} us {b['frac']:.3f}  plain {b.get('plain_order_avg_launch_us', 0):.1f}  copy {b.get('copy_ceiling_GBs', 0):.0f} GB/s x{b.get('frac_of_copy_ceiling', 0):.3f}")
    if lf:
        print(f"    fit floor {lf.get('floor_us')}  rate {lf.get('streaming_GBs')}  resid {lf.get('max_residual_us')}  valid {lf.get('valid')}  {lf.get('avg_launch_us')}")
    if r.get("valu"):
        print(f"    valu {r['valu']['valu_inst_per_env_step']:.1f}/env-step frac {r['valu']['frac']:.3f}  hbm view {r['hbm_view']['frac']:.3f}")
    if r.get("xcd_map"):
        print("    xcd_map", {k: v for k, v in r["xcd_map"].items() if k != "note"})
    c = d.get("cpu_baseline")
    if c:
        print(f"    cpu {c['value'] / 1e6:.0f} M ({c['cores']} threads)  scalar {c['scalar_per_drone']['value'] / 1e6:.0f} M  numpy {c['numpy_vectorised']['value'] / 1e6:.1f} M")
