"""Host-side harness with the reference's entry points (Args, init_model, collate_fn, train/evaluate/test, multi-step
inference, output-directory and checkpoint conventions) driving the HIP hot path.  See each module for the reference
lines it mirrors."""
