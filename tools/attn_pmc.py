"""Launch the attention kernel a few times on the UNet's shapes (eager, one launch each) so that
`rocprofv3 --pmc <counters> --kernel-trace -- python tools/attn_pmc.py` attributes counters per dispatch.
`python tools/attn_pmc.py --summarise <dir>` folds the counter_collection CSVs into a table."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from latentblending_amd.hip import ops as o, lib as l
    DEV = "cuda"
    B = 17
    l.api.lb_attn_set_tuning(int(os.environ.get("LB_ATTN_FORCE", "0")))       # (round 6: 0 = streaming kernel, 513 / 514 = ping-pong form, 256 = rounds 1-5)
    shapes = [(10, 1024, "self"), (20, 256, "self"), (10, 1024, "cross"), (20, 256, "cross")]
    if os.environ.get("LB_ATTN_SELF_ONLY"):
        shapes = shapes[:2]
    for (H, S, kind) in shapes:
        Cc = H * 64
        if kind == "self":
            qkv = torch.randn(B * S, 3 * Cc, device=DEV).half()
            q, k, v, Skv, valid = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], S, S
        else:
            q = torch.randn(B * S, Cc, device=DEV).half()
            kv = torch.randn(B * 80, 2 * Cc, device=DEV).half()
            k, v, Skv, valid = kv[:, :Cc], kv[:, Cc:], 80, 77
        outb = torch.empty(B * S, Cc, device=DEV, dtype=torch.float16)
        for _ in range(3):
            o.attention_d64(q, k, v, B, H, S, Skv, valid, out=outb)
        torch.cuda.synchronize()


def summarise(root):
    rows = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "attn_fwd" not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])
                rows.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for (name, grid), ctr in sorted(rows.items()):
        out[f"{name} grid={grid}"] = {c: sum(v) / len(v) for c, v in sorted(ctr.items())}
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    if "--summarise" in sys.argv:
        res = summarise(sys.argv[sys.argv.index("--summarise") + 1])
        json.dump(res, open(os.path.join(sys.argv[sys.argv.index("--summarise") + 1], "summary.json"), "w"), indent=1)
    else:
        run()
