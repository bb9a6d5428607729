"""The drop-in surface: subclasses of dynesty's own ``Bound`` and
``InternalSampler`` classes backed by the device, to be passed to an
unmodified ``dynesty.NestedSampler`` / ``DynamicNestedSampler`` as

    NestedSampler(prob.loglikelihood, prob.prior_transform, ndim,
                  bound=HipMultiEllipsoid(ndim),
                  sample=HipRWalkSampler(problem=prob),
                  pool=HipBatchPool(queue_size=2000))

Importing this module imports ``dynesty`` (it is the only module of the package
that does).  The sampler classes subclass the *matching* reference classes so
that dynesty's defaults keyed on ``isinstance`` (bootstrap/enlarge:
dynesty.py:186-193; the ncdim restriction: dynesty.py:507-509; update interval:
internal_samplers.py:495-502) apply unchanged, and inherit ``tune`` /
``tune_slice`` (scalar host arithmetic) as is.
"""
from dynesty import bounding as _db
from dynesty import internal_samplers as _dis

from . import bounding as _hb
from . import samplers as _hs
from .pool import HipBatchPool  # noqa: F401  (re-export)


class HipEllipsoid(_hb.HipEllipsoid, _db.Bound):
    """bound=HipEllipsoid(ndim): device twin of dynesty.bounding.Ellipsoid."""


class HipMultiEllipsoid(_hb.HipMultiEllipsoid, _db.Bound):
    """bound=HipMultiEllipsoid(ndim): device twin of
    dynesty.bounding.MultiEllipsoid."""


class HipRadFriends(_hb.HipRadFriends, _db.Bound):
    """bound=HipRadFriends(ndim): device twin of dynesty.bounding.RadFriends
    (bound='balls')."""


class HipSupFriends(_hb.HipSupFriends, _db.Bound):
    """bound=HipSupFriends(ndim): device twin of dynesty.bounding.SupFriends
    (bound='cubes')."""


class _ProblemMixin:

    def _attach_problem(self, kwargs):
        # everything `sample` needs must travel in sampler_kwargs: it is static
        # (internal_samplers.py:161-163)
        self.sampler_kwargs['problem'] = kwargs.get('problem')


class HipRWalkSampler(_ProblemMixin, _dis.RWalkSampler):
    """sample=HipRWalkSampler(problem=..., walks=...)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._attach_problem(kwargs)

    def prepare_sampler(self, loglstar=None, points=None, axes=None, seeds=None, prior_transform=None,
                        loglikelihood=None, nested_sampler=None):
        """internal_samplers.py:111-159, plus per walker the ln L the run holds for its start point
        (`kwargs['logl0']`): what an unmoved walker hands back (samplers.run_rwalk)."""
        args = super().prepare_sampler(loglstar=loglstar, points=points, axes=axes, seeds=seeds,
                                       prior_transform=prior_transform, loglikelihood=loglikelihood,
                                       nested_sampler=nested_sampler)
        logl0 = _hs.stored_logl_of(points, nested_sampler)
        if logl0 is None:
            return args
        return [a._replace(kwargs=dict(a.kwargs, logl0=float(l0))) for a, l0 in zip(args, logl0)]

    sample = staticmethod(_hs.batched(_hs.run_rwalk))


class HipRSliceSampler(_ProblemMixin, _dis.RSliceSampler):
    """sample=HipRSliceSampler(problem=..., slices=...)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._attach_problem(kwargs)

    sample = staticmethod(_hs.batched(_hs.run_rslice))


class HipSliceSampler(_ProblemMixin, _dis.SliceSampler):
    """sample=HipSliceSampler(problem=..., slices=...)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._attach_problem(kwargs)

    sample = staticmethod(_hs.batched(_hs.run_slice))


class HipUniformBoundSampler(_ProblemMixin, _dis.UniformBoundSampler):
    """sample=HipUniformBoundSampler(problem=...); works with an ellipsoidal
    bound (ours or the reference's 'single' / 'multi') and with balls / cubes
    (HipRadFriends / HipSupFriends or the reference's 'balls' / 'cubes')."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._attach_problem(kwargs)

    sample = staticmethod(_hs.batched(_hs.run_unif))
