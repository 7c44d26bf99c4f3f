"""TEST INFRASTRUCTURE ONLY: CPU oracle for the parity tests (see grb_oracle.c)."""
