"""lle_amd -- batched, MI355X-native World.step() for the Laser Learning Environment (drop-in for yamoling/lle's hot path)."""
