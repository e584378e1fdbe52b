"""lycoris_amd -- MI355X (gfx950) native forward/backward for the LyCORIS adapter hot path."""
__version__ = "0.1.0"
