"""CPU oracle for the local-window message-passing path: test infrastructure only."""
