"""Parts of the contract benchmark (bench.py at the repo root is the entry point the driver calls)."""
