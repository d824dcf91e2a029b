"""Test infrastructure only: CPU restatement of the reference algorithm (see cvx_oracle.h)."""
