#!/usr/bin/env python3
"""One rank of a multi-PROCESS shard group (tests/test_gpu_shard_processes.py starts N of these): the control flow of a real deployment —
torch.distributed (gloo) carries the 128-byte unique id, every process owns one libbfhip context and joins the RCCL group with it, the group
proves ONE trace, every process ends up with the whole proof. On a one-GPU box the RCCL entry points come from tests/mock_rccl_ipc.cpp
(BFHIP_RCCL_LIBRARY): real librccl refuses two ranks on one device.

Result: a JSON file per rank {"rank", "proofs": [sha256...], "proof_hex_file", "transport", "error"}; exit code 0 = proved, 3 = the library
reported an error (what a dead peer must lead to), anything else = crashed."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--program", required=True)
    ap.add_argument("--input-hex", default="")
    ap.add_argument("--log-max-rows", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--proofs", type=int, default=1, help="proofs to produce one after the other (the kill test asks for many)")
    ap.add_argument("--conventions", default="0,0,0,0")
    ap.add_argument("--shard-policy", type=int, default=-1, help="bfhip_ctx_set_shard_policy: -1 automatic (two ranks on different GPUs replicate the transforms), 0 exchange, 1 replicate")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = {"rank": rank, "world": world, "pid": os.getpid(), "proofs": []}

    def flush():
        with open(args.out + ".tmp", "w") as f:
            json.dump(out, f)
        os.replace(args.out + ".tmp", args.out)

    flush()
    import importlib.util
    import torch.distributed as dist
    from conftest import load_package
    pkg = load_package()
    spec = importlib.util.spec_from_file_location("replicas", os.path.join(ROOT, "stwo-brainfuck_amd", "replicas.py"))
    replicas = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(replicas)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code = open(args.program).read()
    inp = bytes.fromhex(args.input_hex)
    conv = tuple(int(v) for v in args.conventions.split(","))
    ctx = pkg.Context(args.device, max_log_domain=args.log_max_rows + 2)
    rc = 0
    try:
        ctx.set_conventions(*conv)
        ctx.set_shard_policy(args.shard_policy)
        uid = replicas.share_unique_id(dist, pkg.rccl_unique_id)
        ctx.join_rccl_group(uid, rank, world)
        out["transport"] = ctx.group_info()[2]
        trace = pkg.Trace(ctx, code, inp)
        try:
            for k in range(args.proofs):
                proof, _ = trace.prove(args.log_max_rows)
                out["proofs"].append(hashlib.sha256(proof).hexdigest())
                if k == 0:
                    with open(args.out + ".proof", "wb") as f:
                        f.write(proof)
                out["last_proof_at"] = time.time()
                flush()
            out["group_stats"] = ctx.group_stats()
        finally:
            trace.close()
        ctx.leave_group()
    except pkg.BfhipError as e:
        out["error"] = str(e)
        out["error_at"] = time.time()
        rc = 3
    finally:
        flush()
        try:
            ctx.close()
        except Exception:
            pass
    if rc == 0:
        dist.barrier()
        dist.destroy_process_group()
    # a failed group: the peers may be gone, so no collective teardown — just leave
    sys.stdout.flush()
    os._exit(rc)


if __name__ == "__main__":
    main()
