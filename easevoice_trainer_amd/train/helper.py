"""Output directory naming of the reference (src/train/helper.py:13-69)."""
import datetime
import os
from dataclasses import dataclass
from typing import Optional

train_logs_path = "logs"


def generate_random_name():
    """timestamp name of a run without `output_model_name` (src/train/helper.py:13-15).  Ranks started by dist.spawn_ranks
    share the launcher's stamp (EVT_RUN_STAMP): each would otherwise take its own clock reading, and ranks that start on
    different seconds would write to -- and look for resume checkpoints in -- different directories."""
    return os.environ.get("EVT_RUN_STAMP") or datetime.datetime.now().strftime("%Y%m%d-%H%M%S")


def get_gpt_train_dir(project_dir: str, name: Optional[str]):
    if not name:
        name = "gpt_" + generate_random_name()
    return os.path.join(project_dir, "models", "gpt_train", name)


def get_sovits_train_dir(project_dir: str, name: Optional[str]):
    if not name:
        name = "sovits_" + generate_random_name()
    return os.path.join(project_dir, "models", "sovits_train", name)


@dataclass
class TrainOutput:
    model_path: str


def repo_root():
    return os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# names below <models>/pretrained of the reference (src/utils/config/__init__.py:23-35)
PRETRAINED_FILES = {
    "s2G": os.path.join("gsv-v2final-pretrained", "s2G2333k.pth"),
    "s2D": os.path.join("gsv-v2final-pretrained", "s2D2333k.pth"),
    "s1": os.path.join("gsv-v2final-pretrained", "s1bert25hz-5kh-longer-epoch=12-step=369668.ckpt"),
}


def default_pretrained(kind: str) -> str:
    """The pretrained checkpoint the reference falls back to when a params file leaves the path empty
    (src/train/sovits.py:149-157, the default of GPTTrainParams.model_path): <base>/models/pretrained/gsv-v2final-pretrained/..
    with <base> = the reference checkout.  Looked up under $EVT_PRETRAINED_DIR, then ./models/pretrained (the service starts
    the trainer with cwd = its checkout, src/service/session.py:234-245), then this repository's models/pretrained.
    Returns "" when the file is nowhere: training then starts from scratch, as the reference does when its file is missing
    (sovits.py:345-366 only loads what exists)."""
    roots = [os.environ.get("EVT_PRETRAINED_DIR"), os.path.join(os.getcwd(), "models", "pretrained"),
             os.path.join(repo_root(), "models", "pretrained")]
    for r in roots:
        if r:
            p = os.path.join(r, PRETRAINED_FILES[kind])
            if os.path.isfile(p):
                return p
    return ""
