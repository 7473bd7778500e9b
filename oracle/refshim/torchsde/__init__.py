"""Empty stub of torchsde (only the EDM SDE samplers of the reference use it; out of scope)."""
