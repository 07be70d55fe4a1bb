"""Small functional helpers on the hot path (host side)."""


def broadcast_dim(x):
    """(L,) -> (1,1,L); (B,L) -> (B,1,L); (B,1,L) unchanged; anything else ``ValueError``
    (reference: nnAudio/utils.py:206-222)."""
    if x.dim() == 2:
        x = x[:, None, :]
    elif x.dim() == 1:
        x = x[None, None, :]
    elif x.dim() == 3:
        pass
    else:
        raise ValueError("Only support input with shape = (batch, len) or shape = (len)")
    return x


class ParameterError(NameError):
    """Raised for invalid MFCC parameters.  The reference raises ``ParameterError`` without ever
    defining it (mel.py:255, 273), i.e. the caller sees a ``NameError``; deriving from it keeps
    ``except NameError`` handlers working while giving the error a meaningful name."""
