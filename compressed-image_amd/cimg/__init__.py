"""Python-side plumbing for the MI355X chunk codec engine (ctypes binding + synthetic inputs)."""
