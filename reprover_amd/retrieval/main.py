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


def save_fit_checkpoint(model: PremiseRetriever, ckpt_dir: str, loop: Optional[Dict[str, int]] = None) -> str:
    """What Lightning's ModelCheckpoint keeps for the reference, in this engine's forms: ``<ckpt_dir>/`` is a HuggingFace
    checkpoint directory of the CURRENT weights (``load_hf`` / ``transformers`` read it back),
    ``<ckpt_dir>/training_state.safetensors`` the flat fp32 masters, both AdamW moments, the step counter and the dropout
    stream (``HipT5Trainer.load_training_state`` resumes from it) and ``<ckpt_dir>/loop_state.json`` the position of the
    training loop (epoch, batches of it already trained on, the epoch's seed).  Written to ``<ckpt_dir>.tmp`` and renamed
    into place: a crash in the middle leaves the previous checkpoint (or ``<ckpt_dir>.old``) intact."""
    import json
    import shutil

    tmp, old = ckpt_dir.rstrip("/") + ".tmp", ckpt_dir.rstrip("/") + ".old"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    model.encoder.save_pretrained(tmp)
    model.train_engine().save_training_state(os.path.join(tmp, "training_state.safetensors"))
    with open(os.path.join(tmp, "loop_state.json"), "w") as fh:
        json.dump(loop or {}, fh)
    shutil.rmtree(old, ignore_errors=True)
    if os.path.exists(ckpt_dir):
        os.replace(ckpt_dir, old)
    os.replace(tmp, ckpt_dir)
    shutil.rmtree(old, ignore_errors=True)
    return ckpt_dir


def _epoch_seed(base_seed: int, epoch: int) -> int:
    return (int(base_seed) * 1_000_003 + 7919 * int(epoch) + 12345) % (2 ** 63)


def run_fit(model: PremiseRetriever, dm: RetrievalDataModule, max_steps: int, val_every: int = 0, log=print,
            ckpt_dir: Optional[str] = None, ckpt_every: int = 0, resume_from: Optional[str] = None,
            seed: int = 3407) -> Dict[str, Any]:
    """The training loop Lightning runs for the reference (model.py:146-181): on_fit_start, then per batch
    training_step (forward + backward) → optimizer.step (clipping, AdamW) → scheduler.step → on_train_batch_end;
    epochs until ``max_steps``; validation every ``val_every`` steps (0: never).  With ``ckpt_dir`` the weights and the
    optimizer state are written there at the end (and every ``ckpt_every`` steps); ``resume_from`` = such a directory.

    Every epoch's shuffle and negative sampling draw from ``random`` re-seeded with a function of (``seed``, epoch), and
    a checkpoint records (epoch, batches done): a resumed run re-seeds the epoch and consumes the batches already trained
    on without stepping - the data continues where it stopped, as Lightning restores its loop position, instead of
    replaying the start of the epoch."""
    import json
    import random

    if getattr(dm, "batch_size", 1) <= 0:
        raise ValueError(f"fit needs data.batch_size > 0 (got {dm.batch_size})")
    if dm.ds_train is None:
        dm.setup("fit")
    model.on_fit_start(dm.corpus)
    opt = model.configure_optimizers()
    optimizer, scheduler = opt["optimizer"], opt["lr_scheduler"]["scheduler"]
    step, losses = 0, []
    epoch, skip = 0, 0
    if resume_from:
        if not os.path.isdir(resume_from) and os.path.isdir(resume_from.rstrip("/") + ".old"):
            # a crash between the two renames of save_fit_checkpoint leaves only <dir>.old (+ <dir>.tmp): the last
            # COMPLETE checkpoint is the .old one
            log(f"fit: {resume_from} is missing, resuming from {resume_from.rstrip('/')}.old")
            resume_from = resume_from.rstrip("/") + ".old"
        model.train_engine().load_training_state(os.path.join(resume_from, "training_state.safetensors"))
        step = model.train_engine().steps
        lpath = os.path.join(resume_from, "loop_state.json")
        if os.path.exists(lpath):
            st = json.load(open(lpath))
            epoch, skip = int(st.get("epoch", 0)), int(st.get("batches_done", 0))
            if int(st.get("seed", seed)) != int(seed):
                log(f"fit: the checkpoint's data seed {st['seed']} overrides the configured {seed} "
                    "(the resumed epoch must redraw the batches it already trained on)")
            seed = int(st.get("seed", seed))
    caller_rng = random.getstate()  # the loop re-seeds the global `random` per epoch (the dataset draws from it, as upstream)
    try:
        return _fit_loop(model, dm, max_steps, val_every, log, ckpt_dir, ckpt_every, seed, optimizer, scheduler,
                         step, losses, epoch, skip)
    finally:
        random.setstate(caller_rng)  # ... and hands the caller's stream back untouched


def _fit_loop(model, dm, max_steps, val_every, log, ckpt_dir, ckpt_every, seed, optimizer, scheduler, step, losses,
              epoch, skip) -> Dict[str, Any]:
    import random

    while step < max_steps:
        random.seed(_epoch_seed(seed, epoch))
        # batches trained on before the checkpoint: their examples are drawn again (the same draws) at the index level -
        # not collated, not tokenised, not stepped again (a late-epoch resume cost O(skip) tokenised batches)
        n_epoch = skip
        for batch in dm.train_dataloader(skip=skip):
            n_epoch += 1
            loss = model.training_step(batch, step)
            optimizer.step()
            scheduler.step()
            model.on_train_batch_end(loss, batch, step)
            losses.append(loss)
            step += 1
            if val_every and step % val_every == 0:
                log(f"step {step}: {run_validate(model, dm)}")
            if ckpt_dir and ckpt_every and step % ckpt_every == 0 and step < max_steps:
                rng_state = random.getstate()  # (validation above may draw nothing; the checkpoint itself must not)
                save_fit_checkpoint(model, ckpt_dir, {"epoch": epoch, "batches_done": n_epoch, "seed": seed, "step": step})
                random.setstate(rng_state)
            if step >= max_steps:
                break
        if n_epoch == 0:
            raise ValueError("the training split yields no full batch (drop_last=True)")
        if step < max_steps:
            epoch, skip = epoch + 1, 0
        else:
            skip = n_epoch
    if ckpt_dir:
        save_fit_checkpoint(model, ckpt_dir, {"epoch": epoch, "batches_done": skip, "seed": seed, "step": step})
    return {"steps": step, "losses": [float(x) for x in losses], "checkpoint": ckpt_dir}


def run_validate(model: PremiseRetriever, dm: RetrievalDataModule) -> Dict[str, Any]:
    """on_validation_start + validation_step over the val split (model.py:212-268); returns the
    epoch-level Recall@k (k = 1..num_retrieved, in %) and MRR, weighted by examples with premises as
    the reference's ``self.log(..., batch_size=num_with_premises)`` does."""
    if dm.ds_val is None:
        dm.setup("validate")
    if model.corpus is not dm.corpus:
        model.load_corpus(dm.corpus)
    model.on_validation_start(dm.eval_batch_size)
    for i, batch in enumerate(dm.val_dataloader()):
        model.validation_step(batch, i)
    out = model.epoch_metrics()
    k = model.num_retrieved
    for j in range(k):
        out.setdefault(f"Recall@{j + 1}_val", 0.0)
    out.setdefault("MRR", 0.0)
    out["num_with_premises"] = int(model.logged_metrics.get("MRR", [0.0, 0.0])[1])
    return out


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Premise retriever: fit / predict / validate on MI355X.")
    ap.add_argument("subcommand", choices=["fit", "predict", "validate"])
    ap.add_argument("--max-steps", type=int, default=None, help="fit: overrides trainer.max_steps")
    ap.add_argument("--val-every", type=int, default=0, help="fit: validate every N steps (0: never)")
    ap.add_argument("--config", required=True, help="YAML with `model:` and `data:` sections (reference layout)")
    ap.add_argument("--ckpt_path", default=None, help="HF checkpoint dir (overrides model.model_name)")
    ap.add_argument("--log-dir", default=None, help="where predictions.pickle goes (trainer.log_dir upstream); fit: the "
                                                    "checkpoint is written to <log-dir>/checkpoint")
    ap.add_argument("--ckpt-every", type=int, default=0, help="fit: also checkpoint every N steps (0: only at the end)")
    ap.add_argument("--resume-from", default=None, help="fit: a checkpoint directory written by an earlier fit")
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
            model.dropout_seed = int(seed)  # different seeds, different dropout masks
        if int(d.get("batch_size", 0)) <= 0:
            raise SystemExit("fit: the config's data.batch_size must be a positive integer")
        # Lightning's Trainer defaults default_root_dir to the working directory and always keeps a checkpoint under it
        # (<cwd>/lightning_logs/version_N/checkpoints): a run from a reference-style YAML without a root dir must not
        # train for max_steps and then discard the weights
        log_dir = args.log_dir or tcfg.get("default_root_dir") or os.path.join(os.getcwd(), "lightning_logs")
        print(f"fit: checkpoints go to {os.path.join(log_dir, 'checkpoint')}", flush=True)
        out = run_fit(model, dm, args.max_steps or int(tcfg.get("max_steps", 1)), args.val_every,
                      ckpt_dir=os.path.join(log_dir, "checkpoint"), ckpt_every=args.ckpt_every,
                      resume_from=args.resume_from, seed=int(seed) if seed is not None else 3407)
        first = f"{out['losses'][0]:.6f} -> {out['losses'][-1]:.6f}" if out["losses"] else "(no step taken)"
        print(f"fit: {out['steps']} steps, loss {first}; checkpoint {out['checkpoint']}")
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
