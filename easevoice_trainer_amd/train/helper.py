"""Output directory naming of the reference (src/train/helper.py:13-69)."""
import datetime
import os
from dataclasses import dataclass
from typing import Optional

train_logs_path = "logs"


def generate_random_name():
    return datetime.datetime.now().strftime("%Y%m%d-%H%M%S")


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
