"""CPU oracle (test infrastructure).  See combo_oracle.py's header: never imported by the product package."""
