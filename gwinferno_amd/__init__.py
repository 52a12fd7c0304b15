"""gwinferno_amd -- MI355X-native engine for GWInferno's hierarchical population likelihood
hot path (value, gradient and diagnostic sites of ``hierarchical_likelihood``), behind the
reference's model-callable conventions.  See DESIGN.md."""

__version__ = "0.1.0"
