"""Dataset loop of the reference's evaluation (eval/evaluation_scripts/utils.py:176-266, `infer_dataset`) on the MI355X path:
rank-strided batches (utils.py:181-182), generate → parse → vl_decode through the throughput runner, the caller-side
post-processing (utils.py:252-266) and the two JSONL files with the reference's names and record schema:

    {datasetname}_{rank}_pred_comp_{suffix}.json      {"image_id", "completion"}                      one line per sample
    {datasetname}_{rank}_pred_results_{suffix}.json   {"image_id", "score", "category", "bbox", "mask"} one line per object

Tokenisation / chat templating is the caller's (`prepare`): it needs the checkpoint's tokenizer files, which are not part of
this repository.  `prepare(samples) -> dict(input_ids (local VRT ids), attention_mask, pixel_values, image_grid_thw,
image_sizes [(w, h)], ids [image_id])`.
"""
import json
import os
from typing import Callable, Sequence

from .pipeline import PipelinedRunner, rank_batches
from .postprocess import postprocess_results


def infer_dataset(model, processor, dataset: Sequence, prepare: Callable, output_dir: str, batch_size: int = 1,
                  datasetname: str = "coco", suffix: str = "", rank: int = 0, world: int = 1, max_new_tokens: int = 1024,
                  depth: int = 2, merge: int = 4, schedule=None, repetition_penalty=None, eos_token_id=None):
    """repetition_penalty / eos_token_id: None = the checkpoint's generation_config.json (model.generation_config), as HF's
    generate() inside the reference's loop applies it (utils.py:230-236 → padt.py:436)."""
    os.makedirs(output_dir, exist_ok=True)
    f_res = os.path.join(output_dir, f"{datasetname}_{rank}_pred_results_{suffix}.json")
    f_comp = os.path.join(output_dir, f"{datasetname}_{rank}_pred_comp_{suffix}.json")
    for f in (f_res, f_comp):
        open(f, "w").close()                                      # utils.py:184-188: truncate
    runner = PipelinedRunner(model, processor, depth=depth, merge=merge)
    meta = []                                                     # per submitted batch, in submission (= completion) order

    def drain(finished):
        for decoded, completions, labels, vrts in finished:
            m = meta.pop(0)
            with open(f_comp, "a") as fc:
                for i, completion in enumerate(completions):
                    fc.write(json.dumps({"image_id": m["ids"][i],
                                         "completion": completion.replace("<|endoftext|>", "").replace("<|im_end|>", "")}) + "\n")
            if decoded["pred_boxes"].shape[0] == 0:               # utils.py:253-254
                continue
            with open(f_res, "a") as fr:
                for r in postprocess_results(decoded, labels, m["image_sizes"]):
                    fr.write(json.dumps({"image_id": m["ids"][r["sample_idx"]], "score": r["score"], "category": r["category"],
                                         "bbox": list(r["bbox"]), "mask": r.get("rle")}) + "\n")

    n = 0
    for idx in rank_batches(len(dataset), batch_size, rank, world):
        if idx >= len(dataset):                                   # utils.py:196: ranks past the end skip the work
            continue
        b = prepare(dataset[idx: idx + batch_size])
        meta.append({"ids": list(b["ids"]), "image_sizes": list(b["image_sizes"])})
        drain(runner.submit(b["input_ids"], b["attention_mask"], b["pixel_values"], b["image_grid_thw"],
                            max_new_tokens=max_new_tokens, schedule=schedule, repetition_penalty=repetition_penalty,
                            eos_token_id=eos_token_id))
        n += len(b["ids"])
    drain(runner.flush())
    return {"samples": n, "results_file": f_res, "completions_file": f_comp}
