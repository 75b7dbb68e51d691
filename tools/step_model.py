"""profiles/r03_step_cycles.txt (tools/stepcycles.py on a FOVPT_V_CYCLES=1 build: s_memtime stamps in a sample of the waves) and
profiles/r03_wave_timeline_c3.txt (FOVPT_V_CYCLES=3: every wave's start and end, nothing else) -> profiles/r03_step_model.json:
the latency model of a k_traverse launch that bench.py puts next to the measured launch times.

    launch time  =  (node steps per wave x cycles per node step + leaf steps per wave x cycles per leaf step)   [a wave's chain]
                    / (share of a wave's life spent in steps)  / clock  / (wave slots busy over the launch)     [ramp + tail]

Stamp cost: a sampled wave executes 5 stamps per node step and 4 per leaf step at ~40 cycles each; they are taken out of the
`gap` term (where the bookkeeping sits) to give the step of an unstamped wave."""
import json, re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cyc = open(os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r03_step_cycles.txt")).read()
tl = open(os.path.join(ROOT, "profiles", sys.argv[2] if len(sys.argv) > 2 else "r03_wave_timeline_c3.txt")).read()
out = {"source": "tools/step_model.py from profiles/%s (stamped sample of waves) and profiles/%s (unstamped build)" % (
           sys.argv[1] if len(sys.argv) > 1 else "r03_step_cycles.txt", sys.argv[2] if len(sys.argv) > 2 else "r03_wave_timeline_c3.txt"),
       "launches": {}}
blocks = re.split(r"\n(?=(?:closest|any-hit) it \d)", cyc)
for b in blocks:
    m = re.match(r"(closest|any-hit) it (\d) \| sampled waves\s+(\d+)\s+clock ([\d.]+) GHz\s+wave life\s+(\d+) cyc \(([\d.]+) us\)\s+in steps\s+([\d.]+) %\s+stamp\s+(\d+) cyc", b)
    if not m:
        continue
    kind, it, clock, life, in_steps, stamp = m.group(1), int(m.group(2)), float(m.group(4)), float(m.group(5)), float(m.group(7)) / 100, float(m.group(8))
    n = re.search(r"node steps/wave\s+([\d.]+)\s+cyc/step\s+(\d+) = gap\s+(\d+) \+ load\s+(\d+) \+ alu\s+(\d+) \+ lds\s+(\d+)", b)
    l = re.search(r"leaf steps/wave\s+([\d.]+)\s+cyc/step\s+(\d+) = gap\s+(\d+) \+ load\s+(\d+) \+ rest\s+(\d+)", b)
    d = {"clock_ghz": clock, "node_steps_per_wave": float(n.group(1)), "stamp_cycles": stamp,
         "node_step_cycles_stamped": {"gap": int(n.group(3)), "load": int(n.group(4)), "alu": int(n.group(5)), "lds": int(n.group(6))},
         "leaf_steps_per_wave": float(l.group(1)) if l else 0.0,
         "leaf_step_cycles_stamped": {"gap": int(l.group(3)), "load": int(l.group(4)), "rest": int(l.group(5))} if l else None,
         "share_of_wave_life_in_steps": in_steps}
    # an unstamped wave: 5 stamps per node step (4 in the step, 1 for the hand-over), 4 per leaf step; each segment holds one
    ns = d["node_step_cycles_stamped"]
    d["node_step_cycles"] = {"loop": max(0, ns["gap"] - 2 * stamp), "load_wait": ns["load"] - stamp, "box_test_and_rank": ns["alu"] - stamp, "lds_push_pop": ns["lds"] - stamp}
    d["node_step_cycles"]["total"] = sum(d["node_step_cycles"].values())
    if l:
        ls = d["leaf_step_cycles_stamped"]
        d["leaf_step_cycles"] = {"loop": max(0, ls["gap"] - 2 * stamp), "load_wait": ls["load"] - stamp, "triangle_test_and_pop": ls["rest"] - stamp}
        d["leaf_step_cycles"]["total"] = sum(d["leaf_step_cycles"].values())
    out["launches"]["%s_%d" % (kind, it)] = d
for b in re.split(r"\n(?=(?:closest|any-hit) it \d)", tl):
    m = re.match(r"(closest|any-hit) it (\d)", b)
    t = re.search(r"launch: (\d+) waves, first start -> last end ([\d.]+) us; wave start mean ([\d.]+) .*? wave end mean ([\d.]+) p10 ([\d.]+) p50 ([\d.]+) p90 ([\d.]+) max ([\d.]+) us; life mean ([\d.]+) us; wave slots busy ([\d.]+) %", b)
    if not (m and t):
        continue
    key = "%s_%d" % (m.group(1), int(m.group(2)))
    d = out["launches"].setdefault(key, {})
    d["unstamped"] = {"waves": int(t.group(1)), "launch_us": float(t.group(2)), "wave_end_mean_us": float(t.group(4)), "wave_end_p90_us": float(t.group(7)),
                      "wave_life_mean_us": float(t.group(9)), "wave_slots_busy": float(t.group(10)) / 100}
for key, d in out["launches"].items():
    if "node_step_cycles" in d and "unstamped" in d:
        chain = d["node_steps_per_wave"] * d["node_step_cycles"]["total"] + d["leaf_steps_per_wave"] * d.get("leaf_step_cycles", {"total": 0})["total"]
        life_us = chain / d["share_of_wave_life_in_steps"] / (d["clock_ghz"] * 1e3)
        rounds = max(1.0, d["unstamped"]["waves"] / 8192.0)      # one-wave workgroups: waves per wave slot
        d["model"] = {"chain_cycles_per_wave": round(chain), "wave_life_us": round(life_us, 1), "launch_us": round(life_us * rounds / d["unstamped"]["wave_slots_busy"], 1),
                      "measured_wave_life_us": d["unstamped"]["wave_life_mean_us"], "measured_launch_us": d["unstamped"]["launch_us"],
                      "load_wait_share_of_node_step": round(d["node_step_cycles"]["load_wait"] / d["node_step_cycles"]["total"], 3)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_step_model.json"), "w"), indent=1)
for key, d in out["launches"].items():
    if "model" in d:
        print(key, d["node_step_cycles"], d["model"])
