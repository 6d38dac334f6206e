"""prints the headline, two-context and sequence figures of a bench.py JSON line.  Usage: python tools/show_seq.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], "| two contexts", (d.get("two_contexts") or {}).get("frames_per_s"), (d.get("two_contexts") or {}).get("ms_per_step"),
      "| input-ready", (d.get("input_ready") or {}).get("frames_per_s"), (d.get("input_ready") or {}).get("ms_per_step"))
s = d.get("sequence")
if s:
    print("sequence: pcie", s["pcie_inclusive"]["frames_per_s"], s["pcie_inclusive"].get("upload_GBps_per_gpu"), "GB/s | resident",
          s["hbm_resident"]["frames_per_s"], "| all-pairs", s["keyframe_allpairs"]["seconds"], "s | end to end", s.get("end_to_end_seconds"),
          "| contexts", s["pcie_inclusive"].get("contexts_per_rank"), s["hbm_resident"].get("contexts_per_rank"))
