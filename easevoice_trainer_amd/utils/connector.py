"""Child-process side of the stdout protocol the reference's service layer parses
(src/utils/helper/connector.py:42-144): `<prefix> <json>` lines."""
import json
from typing import Any, Optional

RESP_PREFIX = "response-of-easevoice"
LOSS_PREFIX = "loss-of-easevoice"
LOG_PREFIX = "log-of-easevoice"
SESSION_PREFIX = "session-data-of-easevoice"


class ResponseStatus:
    SUCCESS = "success"
    FAILED = "failed"


class MultiProcessOutputConnector:
    def _print(self, prefix: str, data: str):
        print(f"{prefix} {data}", flush=True)

    def write_response(self, status: str, message: str, data: Optional[dict] = None, uuid=None):
        self._print(RESP_PREFIX, json.dumps({"status": status, "message": message, "data": data, "uuid": uuid}))

    def write_loss(self, step: int, loss: Any, other: Optional[dict] = None):
        d = {"step": step, "loss": loss}
        if other:
            d.update(other)
        self._print(LOSS_PREFIX, json.dumps(d))

    def write_log(self, log: dict):
        self._print(LOG_PREFIX, json.dumps(log))
