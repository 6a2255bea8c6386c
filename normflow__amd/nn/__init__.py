from ._core import Module_, ModuleList_

from .scalar.modules import ConvAct, LinearAct, SplineNet
from .scalar.convNd import ConvNd, Conv4d
from .scalar.modules_ import DistConvertor_, Identity_, Clone_
from .scalar.modules_ import UnityDistConvertor_, PhaseDistConvertor_
from .scalar.modules_ import Expit_, Logit_, SplineNet_, ScaleNet_, SgnBiasNet_

from .scalar.couplings_ import Coupling_, ShiftCoupling_, AffineCoupling_
from .scalar.couplings_ import RQSplineCoupling_, MultiRQSplineCoupling_

from .scalar.spectral_ import FFTNet_, MeanFieldNet_, PSDBlock_, IPSD, lattice_k2
