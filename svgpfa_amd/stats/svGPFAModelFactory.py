"""Model factory with the reference's constants and call signature
(/root/reference/src/svGPFA/stats/svGPFAModelFactory.py:11-148).  The default option set is accelerated
(point process / exponential link / linear embedding / Cholesky Kzz), with q(u) covariances either as Cholesky
vectors (default) or in the reference's rank-1 + diagonal form (converted on the GPU, autograd-transparent);
the other switches raise, exactly where the reference raises or builds an unusable object."""
from . import svLowerBound

#:
PointProcess = 0
#:
Poisson = 1
#:
Gaussian = 2
#:
LinearEmbedding = 100
#:
ExponentialLink = 1000
#:
NonExponentialLink = 1001
#:
kernelMatrixInvChol = 10000
#:
kernelMatrixInvPInv = 10001
#:
indPointsCovRank1PlusDiag = 100000
#:
indPointsCovChol = 100001


class SVGPFAModelFactory:

    @staticmethod
    def buildModelPyTorch(kernels, conditionalDist=PointProcess, linkFunction=ExponentialLink,
                          embeddingType=LinearEmbedding, kernelMatrixInvMethod=kernelMatrixInvChol,
                          indPointsCovRep=indPointsCovChol, device=None, precision="fp64"):
        """precision (extension, not in the reference's signature): "fp64" (the reference's arithmetic) or "mixed"
        (fp32 spike pass and quadrature ELL with fp64 Kzz / Cholesky / KL: BASELINE config 5)."""
        if conditionalDist != PointProcess:
            raise ValueError("Invalid conditionalDist=%s (only PointProcess is built by the HIP engine)"
                             % repr(conditionalDist))
        if embeddingType != LinearEmbedding:
            raise ValueError("Invalid embeddingType={:s}".format(repr(embeddingType)))
        if linkFunction != ExponentialLink:
            raise ValueError("Invalid linkFunction={:s}".format(repr(linkFunction)))
        if indPointsCovRep not in (indPointsCovChol, indPointsCovRank1PlusDiag):
            raise ValueError("Invalid indPointsCovRep")
        if kernelMatrixInvMethod != kernelMatrixInvChol:
            raise ValueError("Invalid kernelMatrixInvMethod (HIP engine: kernelMatrixInvChol only)")
        model = svLowerBound.SVLowerBound(device=device, rank1_plus_diag=(indPointsCovRep == indPointsCovRank1PlusDiag),
                                          precision=precision)
        model.setKernels(kernels=kernels)
        return model
