"""Stop rules with the reference's names and signature `(i, resid, norm_resid, max_resid, tolerance)`
(reference surface: fasta/stopping.py:6-51).  Pure host scalars -- nothing here touches the device."""

__all__ = ["residual", "norm_residual", "ratio_residual", "hybrid_residual"]


def residual(i, resid, norm_resid, max_resid, tolerance):
    """Stop once the residual ||x1 - x0|| / tau is below tolerance (stopping.py:6-15)."""
    return resid < tolerance


def norm_residual(i, resid, norm_resid, max_resid, tolerance):
    """Stop once the normalised residual is below tolerance (stopping.py:18-27)."""
    return norm_resid < tolerance


def ratio_residual(i, resid, norm_resid, max_resid, tolerance):
    """Stop once residual / largest-residual-so-far is below tolerance (stopping.py:30-39)."""
    return resid / max_resid < tolerance


def hybrid_residual(i, resid, norm_resid, max_resid, tolerance):
    """The reference default: ratio rule OR normalised rule (stopping.py:42-51)."""
    return ratio_residual(i, resid, norm_resid, max_resid, tolerance) or \
        norm_residual(i, resid, norm_resid, max_resid, tolerance)
