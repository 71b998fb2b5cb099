#!/usr/bin/env python3
"""The evaluator fed from disk as N ranks on ONE GPU: `python -m hybridgl_amd.main --real` on a synthetic REFER tree
(hybridgl_amd.synth.write_refer_tree), one process per rank, every rank pinned to its share of the host cores
(hybridgl_amd.dist.pin_rank_to_cores), metric rows exchanged over gloo (RCCL cannot put two ranks on one device).
Shows whether the host side -- launch thread + 4 loader threads per rank on 1/N of the cores -- holds the device's rate
when eight ranks of a node run at once.  Prints one JSON object.

    python tools/evaluator_ranks.py --ranks 8 --images 208 --group 8
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--images", type=int, default=208)
    ap.add_argument("--group", type=int, default=8)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--timeout", type=float, default=1500.0)
    args = ap.parse_args()
    from hybridgl_amd import dist as D, synth
    root = tempfile.mkdtemp(prefix="hgl_refer_")
    try:
        info = synth.write_refer_tree(root, n_images=args.images)
        stats = os.path.join(root, "stats.json")
        argv = [sys.executable, "-m", "hybridgl_amd.main", "--real", "--refer_data_root", root, "--dataset", "refcoco", "--split", "val",
                "--bpe_vocab", os.path.join(root, "bpe.txt.gz"), "--parse_json", os.path.join(root, "parse.json"),
                "--proposal_cap", "64", "--pred_iou_thresh=-1e30", "--stability_score_thresh", "0", "--box_nms_thresh", "2.0",
                "--group", str(args.group), "--workers", str(args.workers), "--result_dir", os.path.join(root, "log"),
                "--stats_json", stats]
        env = {"HYBRIDGL_DIST_BACKEND": "gloo", "PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")}
        t0 = time.perf_counter()
        if args.ranks > 1:
            rc = D.spawn_local_ranks(args.ranks, argv, extra_env=env, timeout=args.timeout)
        else:
            rc = subprocess.run(argv, env={**os.environ, **env}, timeout=args.timeout).returncode
        wall = time.perf_counter() - t0
        out = {"ranks": args.ranks, "rc": rc, "wall_s_incl_model_construction": wall, "tree": info}
        if rc == 0 and os.path.exists(stats):
            st = json.load(open(stats))
            s = st["stats"]
            out.update({"value": info["refs"] / s["seconds_job"], "unit": "images/s (all ranks, one shared GPU)",
                        "seconds_job": s["seconds_job"], "host_cores_per_rank": s["host_cores_per_rank"],
                        "rank0": s, "metrics": st["metrics"]})
        print(json.dumps(out))
        return rc
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
