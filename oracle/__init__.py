"""CPU oracle (test infrastructure only). See oracle/drone_oracle.c."""
