"""``fit`` / ``predict`` / ``validate`` driver of the retriever — stands in for the reference's
``python retrieval/main.py {fit,predict,validate} --config …`` (retrieval/main.py:12-21, a LightningCLI).
Reads the same YAML keys the reference's configs use (``model.{model_name,lr,warmup_steps,num_retrieved}``,
``data.{data_path,corpus_path,num_negatives,num_in_file_negatives,batch_size,eval_batch_size,max_seq_len}``,
``trainer.{max_steps,gradient_clip_val}``; retrieval/confs/*.yaml) and writes ``<log_dir>/predictions.pickle`` exactly
as ``on_predict_epoch_end`` does (model.py:329-336).  Lightning itself is out of scope; the hooks' bodies live in
``model.py`` here as they do upstream, and ``run_fit`` calls them in Lightning's order.

Multi-GPU ``predict``: launch under ``python -m torch.distributed.run --nproc-per-node N -m
reprover_amd.retrieval.main predict ...``.  Every rank walks the same batches; the index is row-sharded
(each rank re-indexes 1/N of the corpus' tokens), per-rank top-k lists are merged through one
all-gather, rank 0 writes ``predictions.pickle``.
"""
from __future__ import annotations

import argparse
import os
from typing import Any, Dict, Optional

import torch
import yaml

from .datamodule import RetrievalDataModule
from .evaluate import recall_and_mrr
from .model import PremiseRetriever


def run_predict(model: PremiseRetriever, dm: RetrievalDataModule, log_dir: Optional[str]) -> int:
    """on_predict_start → predict_step over the predict split → on_predict_epoch_end."""
    dm.setup("predict")
    model.on_predict_start(dm.corpus, dm.eval_batch_size)
    n = 0
    for batch in dm.predict_dataloader():  # host batches go in as they are: the encoder stages them through pinned memory
        model.predict_step(batch, n)
        n += 1
    count = len(model.predict_step_outputs)
    model.on_predict_epoch_end(log_dir)
    return count


def run_fit(model: PremiseRetriever, dm: RetrievalDataModule, max_steps: int, val_every: int = 0, log=print) -> Dict[str, Any]:
    """The training loop Lightning runs for the reference (model.py:146-181): on_fit_start, then per batch
    training_step (forward + backward) → optimizer.step (clipping, AdamW) → scheduler.step → on_train_batch_end;
    epochs until ``max_steps``; validation every ``val_every`` steps (0: never)."""
    if dm.ds_train is None:
        dm.setup("fit")
    model.on_fit_start(dm.corpus)
    opt = model.configure_optimizers()
    optimizer, scheduler = opt["optimizer"], opt["lr_scheduler"]["scheduler"]
    step, losses = 0, []
    while step < max_steps:
        n_epoch = 0
        for batch in dm.train_dataloader():
            loss = model.training_step(batch, step)
            optimizer.step()
            scheduler.step()
            model.on_train_batch_end(loss, batch, step)
            losses.append(loss)
            step += 1
            n_epoch += 1
            if val_every and step % val_every == 0:
                log(f"step {step}: {run_validate(model, dm)}")
            if step >= max_steps:
                break
        if n_epoch == 0:
            raise ValueError("the training split yields no full batch (drop_last=True)")
    return {"steps": step, "losses": [float(x) for x in losses]}


def run_validate(model: PremiseRetriever, dm: RetrievalDataModule) -> Dict[str, Any]:
    """on_validation_start + validation_step over the val split (model.py:212-268); returns the
    epoch-level Recall@k (k = 1..num_retrieved, in %) and MRR, weighted by examples with premises as
    the reference's ``self.log(..., batch_size=num_with_premises)`` does."""
    if dm.ds_val is None:
        dm.setup("validate")
    if model.corpus is not dm.corpus:
        model.load_corpus(dm.corpus)
    model.reindex_corpus(dm.eval_batch_size)
    k = model.num_retrieved
    tot_recall = [0.0] * k
    tot_mrr, tot_n = 0.0, 0
    for batch in dm.val_dataloader():
        emb = model._encode(batch["context_ids"], batch["context_mask"])
        retrieved, _ = model.corpus.get_nearest_premises(model.corpus_embeddings, batch["context"], emb, k)
        if not any(len(p) for p in batch["all_pos_premises"]):
            continue
        recall, mrr, n = recall_and_mrr(batch["all_pos_premises"], retrieved, k)
        for j in range(k):
            tot_recall[j] += recall[j] * n
        tot_mrr += mrr * n
        tot_n += n
    out = {f"Recall@{j + 1}_val": tot_recall[j] / max(tot_n, 1) for j in range(k)}
    out["MRR"] = tot_mrr / max(tot_n, 1)
    out["num_with_premises"] = tot_n
    return out


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Premise retriever: fit / predict / validate on MI355X.")
    ap.add_argument("subcommand", choices=["fit", "predict", "validate"])
    ap.add_argument("--max-steps", type=int, default=None, help="fit: overrides trainer.max_steps")
    ap.add_argument("--val-every", type=int, default=0, help="fit: validate every N steps (0: never)")
    ap.add_argument("--config", required=True, help="YAML with `model:` and `data:` sections (reference layout)")
    ap.add_argument("--ckpt_path", default=None, help="HF checkpoint dir (overrides model.model_name)")
    ap.add_argument("--log-dir", default=None, help="where predictions.pickle goes (trainer.log_dir upstream)")
    args = ap.parse_args(argv)
    with open(args.config) as fh:
        cfg = yaml.safe_load(fh)
    m, d = cfg["model"], cfg["data"]
    if not torch.cuda.is_available():
        raise RuntimeError("reprover_amd needs an MI355X (HIP) device; no CPU fallback exists")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("RP_DIST_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("RP_DIST_BACKEND", "nccl")  # gloo: functional test on a one-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    model = PremiseRetriever.load_hf(args.ckpt_path or m["model_name"], d["max_seq_len"], device)
    model.num_retrieved = m.get("num_retrieved", 100)
    model.shard_index_over_ranks = world > 1 and args.subcommand == "predict"
    dm = RetrievalDataModule(d["data_path"], d["corpus_path"], d["eval_batch_size"], d["max_seq_len"], model.tokenizer,
                             num_negatives=d.get("num_negatives", 0), num_in_file_negatives=d.get("num_in_file_negatives", 0),
                             batch_size=d.get("batch_size", 0))
    if args.subcommand == "fit":
        assert world == 1, "fit runs one process (data-parallel training is not part of this path)"
        tcfg = cfg.get("trainer", {})
        model.lr, model.warmup_steps = float(m.get("lr", 0.0)), int(m.get("warmup_steps", 0))
        model.gradient_clip_val = tcfg.get("gradient_clip_val")
        seed = cfg.get("seed_everything")
        if seed is not None:
            import random

            random.seed(int(seed))
            torch.manual_seed(int(seed))
        out = run_fit(model, dm, args.max_steps or int(tcfg.get("max_steps", 1)), args.val_every)
        print(f"fit: {out['steps']} steps, loss {out['losses'][0]:.6f} -> {out['losses'][-1]:.6f}")
    elif args.subcommand == "predict":
        log_dir = args.log_dir or cfg.get("trainer", {}).get("default_root_dir") or os.getcwd()
        os.makedirs(log_dir, exist_ok=True)
        n = run_predict(model, dm, log_dir if rank == 0 else None)
        if rank == 0:
            print(f"{n} retrieval predictions saved to {os.path.join(log_dir, 'predictions.pickle')}")
    else:
        for k, v in run_validate(model, dm).items():
            if rank == 0:
                print(f"{k}: {v}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
