"""recon_amd — MI355X-native graph-context aggregation for RECON (GAT attention layer + GP-GNN
propagation), behind the reference's own nn.Module surface.  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"
