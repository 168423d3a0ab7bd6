"""The process boundary of SURVEY §8(b), CPU tier: cmd/train_sovits.py / cmd/train_gpt.py started the way the reference's
service starts them (src/service/session.py:234-253: Popen([python, cmd, "-c", tmp.json])) and their stdout read by the
reference's OWN line parser (src/utils/helper/connector.py:42-144) when a checkout is present, by a restatement of it
otherwise.  Without a GPU the trainers cannot run (there is no CPU fallback), so what this tier pins is the contract's
failure half -- exceptions never escape, exit code 0, exactly one `response-of-easevoice {"status": "failed", ...}` line,
everything else passes through as plain lines -- and that every line our connector writes parses on the other side into
the reference's own records.  The success half runs on the GPU box: tests/test_zz_process_boundary_gpu.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def parse_lines(lines):
    """[(kind, payload)] -- through the reference's parser when its checkout is here, else the same rules restated"""
    if os.path.isdir(os.path.join(REF, "src", "utils", "helper")):
        sys.path.insert(0, REF)
        try:
            from src.utils.helper.connector import ConnectorDataType, MultiProcessOutputConnector as RefConn
        finally:
            sys.path.remove(REF)
        rc, out = RefConn(), []
        for l in lines:
            d = rc._parse_result(l.strip())
            assert d is not None, f"the reference's parser rejected <{l}>"
            if d.dataType == ConnectorDataType.RESP:
                out.append(("resp", dict(status=d.response.status, message=d.response.message, data=d.response.data)))
            elif d.dataType == ConnectorDataType.LOSS:
                out.append(("loss", dict(step=d.loss.step, loss=d.loss.loss, other=d.loss.other)))
            elif d.dataType == ConnectorDataType.LOG:
                out.append(("log", d.log))
            elif d.dataType == ConnectorDataType.SESSION_DATA:
                out.append(("session", d.session_data))
            else:
                out.append(("other", d.other))
        return out
    out = []
    for l in lines:
        l = l.strip()
        for kind, pre in (("resp", "response-of-easevoice"), ("loss", "loss-of-easevoice"), ("log", "log-of-easevoice"),
                          ("session", "session-data-of-easevoice")):
            if l.startswith(pre):
                d = json.loads(l[len(pre):].strip())
                if kind == "loss":
                    d = dict(step=d.pop("step"), loss=d.pop("loss"), other=d)
                out.append((kind, d))
                break
        else:
            out.append(("other", l))
    return out


def run_child(script, config, tmp_path, extra_env=None):
    cfg = tmp_path / "params.json"
    cfg.write_text(json.dumps(config))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "cmd", script), "-c", str(cfg)], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    return r


@pytest.mark.parametrize("script,config", [
    ("train_sovits.py", dict(batch_size=2, total_epochs=1, train_input_dir="/nonexistent/dir", output_model_name="x")),
    ("train_gpt.py", dict(batch_size=2, total_epochs=1, train_input_dir="/nonexistent/dir", output_model_name="x")),
    ("train_sovits.py", dict(no_such_field=1)),
    ("train_gpt.py", dict(no_such_field=1)),
])
def test_failed_run_answers_failed_and_exits_zero(script, config, tmp_path):
    config = dict(config)
    if "no_such_field" not in config:
        config["project_dir"] = str(tmp_path)
    r = run_child(script, config, tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]                    # the exit code is ignored by the parent: must not matter
    recs = parse_lines(r.stdout.splitlines())
    resp = [p for k, p in recs if k == "resp"]
    assert len(resp) == 1 and resp[0]["status"] == "failed", r.stdout[-2000:]
    assert resp[0]["message"].startswith("failed to train " + ("sovits" if "sovits" in script else "gpt"))
    assert recs[-1][0] == "resp"                                  # the answer is the last line
    assert "Traceback" in r.stderr                                # the cause goes to stderr, which the parent echoes


def test_every_line_our_connector_writes_parses_on_the_other_side(capsys):
    from easevoice_trainer_amd.utils.connector import MultiProcessOutputConnector, ResponseStatus

    c = MultiProcessOutputConnector()
    c.write_loss(5, 1.25, {"lr": 1e-4, "epoch": 1})
    c.write_loss(10, 0.5)
    c.write_log({"msg": "hello", "n": 3})
    c.write_response(ResponseStatus.SUCCESS, "Finish train sovits", data={"model_path": "/tmp/m"})
    print("a plain line that is none of the four prefixes")
    recs = parse_lines(capsys.readouterr().out.splitlines())
    assert recs[0] == ("loss", dict(step=5, loss=1.25, other={"lr": 1e-4, "epoch": 1}))
    assert recs[1] == ("loss", dict(step=10, loss=0.5, other={}))
    assert recs[2] == ("log", {"msg": "hello", "n": 3})
    assert recs[3][0] == "resp" and recs[3][1]["status"] == "success" and recs[3][1]["data"] == {"model_path": "/tmp/m"}
    assert recs[4] == ("other", "a plain line that is none of the four prefixes")
