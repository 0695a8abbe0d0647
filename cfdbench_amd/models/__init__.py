"""Host-side mirror of the reference's ``src/models`` plugin surface for the accelerated path."""
