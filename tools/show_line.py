"""Print the numbers of a bench.py JSON line that the round tracks (headline, Large sections with their one-stream stages, training)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print("headline", d["value"], d["unit"], d["ms_per_step"], "ms/step")
L = d.get("configs2_large_b16")
if L:
    print("large", L["value"], L["ms_per_step"], "one batch per forward", L["one_batch_per_forward"]["value"], "one stream ms", L["one_stream_ms_per_step"])
    for s in L["stages_one_stream"]:
        print("  ", s["kernel"], s["ms"], s["frac"])
for k in ("configs3_train_b16", "train_large_b16"):
    if k in d:
        print(k, d[k]["ms_per_step"], "ms/step")
if "configs4_corpus" in d:
    c = d["configs4_corpus"]
    print("corpus", c.get("wall_s"), "s", c.get("value"), c.get("unit"))
