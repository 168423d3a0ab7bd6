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
    try:
        parser = argparse.ArgumentParser(description="run train sovits")
        parser.add_argument("-c", "--config", type=argparse.FileType("r"), required=True)
        args = parser.parse_args()
        config = json.loads(args.config.read())
        args.config.close()
        from easevoice_trainer_amd.train.sovits import SovitsTrain, SovitsTrainParams

        train = SovitsTrain(params=SovitsTrainParams(**config))
        output = train.train()
        connector.write_response(ResponseStatus.SUCCESS, "Finish train sovits", data=asdict(output))
    except Exception as e:
        traceback.print_exc()
        connector.write_response(ResponseStatus.FAILED, f"failed to train sovits, {e}")


if __name__ == "__main__":
    main()
