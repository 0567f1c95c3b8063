"""reprover_amd — MI355X-native premise-retrieval engine (drop-in for the hot path of
lean-dojo/ReProver: retrieval/model.py::PremiseRetriever, common.py::Corpus.get_nearest_premises,
retrieval/index.py).  Compute runs in hand-written HIP kernels (libreprover_hip.so, gfx950)
behind the C ABI declared in include/reprover_hip.h; see DESIGN.md and INTEGRATION.md."""

__all__ = ["common", "encoder", "tokenizer", "synth", "retrieval", "build"]
