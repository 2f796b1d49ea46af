"""CPU oracle (test infrastructure only): see oracle/nvfi_oracle.h. Never imported by nvfi_amd."""
