#!/usr/bin/env python3
"""hipBLASLt / rocBLAS solution selection for the encoder's GEMM shapes with PyTorch's TunableOp, on the box it will run on:
the full-size ColPali-v1.2 architecture at the reference worker's 16 pages per forward (and the query path), timings before and
after, the selected solutions written to a CSV the adapters load (morphik-core_amd/tuned/).  PyTorch validates the file against
the ROCm / hipBLASLt versions and the GPU architecture and ignores it on a mismatch.

  python tools/tune_encoder_gemms.py [--batches 16,32] [--out gpurun_out/tunableop_colpali_v1_2.csv]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,32")
    ap.add_argument("--out", default="gpurun_out/tunableop_colpali_v1_2.csv")
    ap.add_argument("--max-ms", type=int, default=30, help="tuning time budget per candidate solution")
    ap.add_argument("--use", default="", help="do not tune: load this file and time the forward with its solutions")
    a = ap.parse_args()
    import torch
    import torch.cuda.tunable as tun

    from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel

    os.environ["MV_ENCODER_TUNED_GEMMS"] = "0"  # the adapter must not load the shipped selections: measure the untuned baseline first
    emb = MI355XColpaliEmbeddingModel(preset="colpali-v1.2", device="cuda:0", batch_size=64)
    rng = np.random.default_rng(3)
    batches = [int(b) for b in a.batches.split(",")]
    imgs = [rng.integers(0, 255, (emb.image_size, emb.image_size, 3), dtype=np.uint8) for _ in range(max(batches))]

    def pages_per_s(b, reps=6):
        emb._embed_images_device(imgs[:b])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            emb._embed_images_device(imgs[:b])
        torch.cuda.synchronize()
        return b * reps / (time.perf_counter() - t0)

    res = {"fused_encoder_ops": emb.fused_ops, "untuned_pages_per_s": {b: round(pages_per_s(b), 2) for b in batches}}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    tun.enable(True)
    if a.use:  # no tuning: load a file produced earlier (what the adapters do)
        tun.tuning_enable(False)
        res["read_ok"] = bool(tun.read_file(a.use))
    else:
        tun.tuning_enable(True)
        tun.set_max_tuning_duration(a.max_ms)
        tun.set_filename(a.out)  # PyTorch writes the selected solutions here when the process exits
        t0 = time.perf_counter()
        for b in batches:
            emb._embed_images_device(imgs[:b])
        emb._embed_texts_device(["total revenue by quarter in the third fiscal year", "hello"])
        torch.cuda.synchronize()
        res["tuning_s"] = round(time.perf_counter() - t0, 1)
        tun.tuning_enable(False)
    res["solutions"] = len(tun.get_results())
    res["tuned_pages_per_s"] = {b: round(pages_per_s(b), 2) for b in batches}
    res["validators"] = [list(v) for v in tun.get_validators()]
    res["file"] = a.out
    print(json.dumps(res))


if __name__ == "__main__":
    main()
