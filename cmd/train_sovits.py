#!/usr/bin/env python
"""Subprocess entry point with the reference's contract (src/cmd/train_sovits.py:20-43): `-c params.json`, progress as
`loss-of-easevoice {...}` lines, final `response-of-easevoice {...}`; exceptions never escape."""
import argparse
import json
import os
import sys
import traceback
from dataclasses import asdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easevoice_trainer_amd.utils.connector import MultiProcessOutputConnector, ResponseStatus  # noqa: E402


def main():
    connector = MultiProcessOutputConnector()
    spawned = os.environ.get("EVT_SPAWNED") == "1"      # one of several workers started by spawn_ranks below
    rank0 = int(os.environ.get("RANK", "0")) == 0
    try:
        parser = argparse.ArgumentParser(description="run train sovits")
        parser.add_argument("-c", "--config", type=argparse.FileType("r"), required=True)
        args = parser.parse_args()
        config = json.loads(args.config.read())
        args.config.close()
        from easevoice_trainer_amd.dist import parse_gpu_ids, spawn_ranks
        from easevoice_trainer_amd.train.sovits import SovitsTrain, SovitsTrainParams

        params = SovitsTrainParams(**config)
        ids = parse_gpu_ids(params.gpu_ids)
        if len(ids) > 1 and "WORLD_SIZE" not in os.environ:
            # several GPUs requested and no launcher around us: one process per GPU; rank 0 answers for the job
            codes = spawn_ranks([sys.executable, os.path.abspath(__file__), "-c", args.config.name], ids)
            if any(c != 0 for c in codes):      # a failing worker prints its traceback and leaves the answer to us
                connector.write_response(ResponseStatus.FAILED, f"failed to train sovits, worker exit codes {codes}")
            return
        output = SovitsTrain(params=params).train()
        if rank0:
            connector.write_response(ResponseStatus.SUCCESS, "Finish train sovits", data=asdict(output))
    except Exception as e:
        traceback.print_exc()
        if spawned:
            sys.exit(1)       # the launcher stops the other ranks and answers FAILED for the job
        connector.write_response(ResponseStatus.FAILED, f"failed to train sovits, {e}")


if __name__ == "__main__":
    main()
