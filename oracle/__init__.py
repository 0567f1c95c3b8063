"""CPU oracle for the premise-retrieval hot path — TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU and in the plainest possible form, what the reference
(lean-dojo/ReProver @ /root/reference, plus the HuggingFace `transformers` T5 encoder it
delegates to) computes on the path named by BASELINE.json:north_star:

  * ``oracle.common_ref``  — Pos / Context / Premise.serialize / File filters / Corpus DAG /
    accessibility / ``get_nearest_premises``            (reference: common.py:34-338)
  * ``oracle.t5_ref``      — ByT5 byte tokenizer, T5 relative-position buckets, the T5 encoder
    forward, masked mean-pool + L2 normalise            (reference: retrieval/model.py:92-114;
    transformers v5.15.0 models/t5/modeling_t5.py, models/byt5/tokenization_byt5.py)

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` — and there only as the checker / the timed CPU baseline.  Nothing under
``reprover_amd/`` imports it; the product path fails loudly when the HIP library is missing.

Parity pinning: the oracle is pinned against golden vectors generated in the authoring container
by importing the reference itself and HuggingFace (``tests/golden/make_golden.py`` →
``tests/golden/*.npz|json``; the reference has no tests or golden vectors of its own for this
path — SURVEY.md §4, §8c).  ``tests/test_oracle_golden.py`` re-checks every fixture on each run.
"""
