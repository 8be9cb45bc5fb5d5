"""What a run/delta encoding of the neighbour lists would save over the fixed 10-bit entries.

Neighbour lists of the dam break from the CPU oracle (ascending sorted-slot indices, neighborhood_search.rs:262-273) at a few points
in time; per particle: entries, runs of consecutive indices, and the bits of four encodings:
  fixed10      10 bits per entry (this build: staging-area slots, three per 32-bit word)
  run12        a run = start:10 | len-1:2  (runs longer than 4 split)
  run13        a run = start:10 | len-1:3  (runs longer than 8 split)
  delta        Band et al. style: first entry 10 bits, then per entry a delta in the smallest of {2, 4, 10} bits + a 2-bit
               width tag per group of 4 (lower bound: no alignment cost)
Usage: python tools/list_run_stats.py [scale] [steps ...]      (scale 0.3 -> ~90 K particles)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import oracle as orc
from util import dam_break

def stats(o):
    counts, start, lists = o.neighbors()
    n = o.n
    k = counts[:, 1].astype(np.int64)
    tot = int(k.sum())
    owner = np.repeat(np.arange(n), k)
    first = np.zeros(tot, bool); first[start[:-1][k > 0].astype(np.int64)] = True
    d = np.diff(lists.astype(np.int64), prepend=-10)
    newrun = first | (d != 1)
    runs = np.bincount(owner[newrun], minlength=n)
    # run lengths
    rid = np.cumsum(newrun) - 1
    rlen = np.bincount(rid)
    run12 = (np.ceil(rlen / 4).sum() * 12)
    run13 = (np.ceil(rlen / 8).sum() * 13)
    dd = np.where(first, 0, d)
    bits = np.where(first, 10, np.where(dd < 4, 2, np.where(dd < 16, 4, 10)))
    delta = bits.sum() + 2 * np.ceil(k / 4).sum()
    return dict(n=n, k=tot / n, runs=runs.mean(), mean_run=rlen.mean(), fixed10=10 * tot / n, run12=run12 / n, run13=run13 / n, delta=delta / n)

def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    steps = [int(a) for a in sys.argv[2:]] or [0, 200, 1000]
    fluid, boundary = dam_break(scale)  # (the scene comes from the host mirror: the library must be built, no GPU needed)
    o = orc.Oracle(); o.set_boundary(np.array(boundary)); o.set_particles(np.array(fluid))
    done = 0
    for s in steps:
        while done < s:
            o.dfsph_step(); done += 1
        if done == 0: o.update_neighborhood()
        r = stats(o)
        print("step %5d  n %d  entries/particle %.2f  runs/particle %.2f  mean run %.2f  bits/particle: fixed10 %.1f  run12 %.1f  run13 %.1f  delta %.1f"
              % (done, r["n"], r["k"], r["runs"], r["mean_run"], r["fixed10"], r["run12"], r["run13"], r["delta"]))

if __name__ == "__main__":
    main()
