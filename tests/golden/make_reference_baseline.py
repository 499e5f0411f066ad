"""Extracts the head of the reference's own golden-loss file (tests/test_baseline.json, the 9 x 100 training losses its
tests/test_tutel.py:94-152 compares helloworld runs against) into tests/golden/reference_baseline_losses.json: per entry the
helloworld flags test_tutel.py:42 passes, and the first HEAD losses verbatim.  Only runs in the build container
(the GPU box has no /root/reference).

    python tests/golden/make_reference_baseline.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TUTEL_REFERENCE", "/root/reference")
HEAD = 24

src = json.load(open(os.path.join(REF, "tests", "test_baseline.json")))
cases = []
for ent in src:
    batch = 1 if ent["dtype"] == "float64" else 16   # test_tutel.py:150-152 runs the fp64 case with batch_size=1
    cases.append({"top": int(ent["top"]), "dtype": ent["dtype"], "num_local_experts": int(ent["num_local_experts"]), "batch_size": batch,
                  "num_tokens": 1024, "model_dim": 2048, "hidden_size": 2048,
                  "flags": "--top %s --dtype %s --num_local_experts %s --hidden_size 2048 --batch_size %d --a2a_ffn_overlap_degree 1 "
                           "--num_tokens 1024 --parallel_type data" % (ent["top"], ent["dtype"], ent["num_local_experts"], batch),
                  "losses": ent["losses"][:HEAD]})
out = {"_comment": "head of /root/reference/tests/test_baseline.json (losses printed by the reference's helloworld on its CI GPUs); "
                   "made by tests/golden/make_reference_baseline.py", "rounding": "test_tutel.py:52-63,79-83: 3 decimals for float32, 1 decimal otherwise",
       "cases": cases}
json.dump(out, open(os.path.join(HERE, "reference_baseline_losses.json"), "w"), indent=1)
print("wrote", len(cases), "cases")
