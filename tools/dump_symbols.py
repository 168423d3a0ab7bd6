"""Writes the phoneme table the feature-directory readers need (easevoice_trainer_amd/train/dataset.py) as a JSON list
whose index is the phoneme id, taken from a reference checkout's text front-end (src/easevoice/text/symbols.py).

    python tools/dump_symbols.py /path/to/easevoice-trainer  <exp_dir>/symbols.json
"""
import json
import sys


def main():
    ref_root, out = sys.argv[1], sys.argv[2]
    sys.path.insert(0, ref_root)
    from src.easevoice.text.symbols import SYMBOLS, SYMBOLS_TO_ID

    assert all(SYMBOLS_TO_ID[s] == i for i, s in enumerate(SYMBOLS))
    with open(out, "w", encoding="utf8") as f:
        json.dump(list(SYMBOLS), f, ensure_ascii=False)
    print(f"{len(SYMBOLS)} symbols -> {out}")


if __name__ == "__main__":
    main()
