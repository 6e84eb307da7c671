"""TEST INFRASTRUCTURE ONLY: the CPU oracle of the DAAM extraction path (numpy restatement pinned to golden vectors
of the unmodified reference), diffusers stand-ins, the golden-vector generator and the torch port used as the timed
CPU / eager-GPU baseline.  Nothing under ``daam_amd/`` imports this package."""
