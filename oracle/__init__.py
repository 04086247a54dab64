"""CPU oracle (test infrastructure only; see nic_oracle.py).  The product package never imports it."""
