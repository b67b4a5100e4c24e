"""Bayesian-quadrature moment transforms (counterpart of the reference's ssmtoybox/bq package)."""
