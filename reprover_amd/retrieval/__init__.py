"""Mirror of the reference's ``retrieval/`` package for the inference hot path."""
