"""Test infrastructure only: CPU restatement of the reference hot path (see edtr_oracle.py)."""
