"""features.jsonl -> tokens.jsonl (`audio_repr` strings): /root/reference cli/prepare_tokens.py:14-54.

  python -m slamkit_amd.cli.prepare_tokens data_path=example_data/features.jsonl out_path=/tmp/out
"""
import json
import logging
import os
import sys

from ..tokeniser import tokeniser_factory
from ..utils.config import load_config


def process_jsonl(line: str, tokeniser, requires_meta: bool = False, meta_path=None):
    """prepare_tokens.py:14-35. `requires_meta` (interleaving tokeniser): merge `<meta_path>/<stem>.json`, or the
    json next to `file_name`, into the row first - it carries `aligned_text` (word, start, end)."""
    try:
        cur = json.loads(line)
        if requires_meta:
            stem = os.path.splitext(os.path.basename(cur["file_name"]))[0]
            meta_file = f"{meta_path}/{stem}" if meta_path else os.path.splitext(cur["file_name"])[0]
            if not os.path.exists(meta_file + ".json"):
                logging.warning(f"{meta_file} does not exist. Skipping")
                return None
            with open(meta_file + ".json") as f:
                cur.update(json.load(f))
        cur["audio_repr"] = tokeniser.stringify_representation([cur], mode="train")[0]
        for k in ("units", "duration", "text", "aligned_text", "split_sentence"):
            cur.pop(k, None)
        return json.dumps(cur)
    except Exception as e:  # noqa: BLE001
        logging.warning(f"Failed to process {line}. Error: {e}, skipping")
        return None


def prepare_tokens(argv=None):
    cfg = load_config("prepare_tokens", list(argv if argv is not None else sys.argv[1:]))
    tokeniser = tokeniser_factory(cfg.tokeniser)
    os.makedirs(cfg.out_path, exist_ok=True)
    out_path = f"{cfg.out_path}/{cfg.data_path.split('/')[-1]}"
    if os.path.exists(out_path):
        logging.warning(f"{out_path} already exists. Deleting it!")
        os.remove(out_path)
    with open(cfg.data_path) as f_in, open(out_path, "a+") as f_out:
        for line in f_in:
            js = process_jsonl(line, tokeniser, bool(cfg.tokeniser.get("requires_meta", False)), cfg.get("meta_path", None))
            if js:
                f_out.write(js + "\n")
    return out_path


if __name__ == "__main__":
    prepare_tokens()
