"""Model factory with the reference's constants and call signature
(/root/reference/src/svGPFA/stats/svGPFAModelFactory.py:11-148).  The default option set is accelerated
(point process / exponential link / linear embedding / Cholesky Kzz), with q(u) covariances either as Cholesky
vectors (default) or in the reference's rank-1 + diagonal form (converted on the GPU, autograd-transparent);
the other switches raise, exactly where the reference raises or builds an unusable object.

kernelMatrixInvPInv (reference stats/kernelsMatricesStore.py:146-165: Kzz^+ by torch.linalg.pinv, rcond 1e-15, applied by
multiplication) is accepted and served by the same engine: for a positive definite Kzz + eps I the pseudo-inverse is the
inverse, which the engine applies in whitened (Cholesky) form -- more accurately than an explicit pseudo-inverse;
tests/golden/pinv_small.npz (the reference's own PInv model) pins value and gradients.  A Kzz that is NOT positive definite
(where the reference's pseudo-inverse would silently project) raises torch.linalg.LinAlgError in both modes."""
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
                          indPointsCovRep=indPointsCovChol, device=None, precision="fp64", linkFunctionCallable=None):
        """precision (extension, not in the reference's signature): "fp64" (the reference's arithmetic) or "mixed"
        (fp32 spike pass and quadrature ELL with fp64 Kzz / Cholesky / KL: BASELINE config 5).
        linkFunction=NonExponentialLink with linkFunctionCallable=g (extension): the point-process ELL with
        Gauss-Hermite quadrature, the reference's PointProcessELLQuad (expectedLogLikelihood.py:215-255) -- a class the
        reference's own factory cannot reach (it raises for every link but the exponential, svGPFAModelFactory.py:96-98)
        and that its tests assemble by hand.  eLLCalculationParams must then hold hermQuadPoints / hermQuadWeights."""
        if conditionalDist != PointProcess:
            raise ValueError("Invalid conditionalDist=%s (only PointProcess is built by the HIP engine)"
                             % repr(conditionalDist))
        if embeddingType != LinearEmbedding:
            raise ValueError("Invalid embeddingType={:s}".format(repr(embeddingType)))
        if linkFunction == NonExponentialLink and linkFunctionCallable is not None:
            if indPointsCovRep != indPointsCovChol or precision != "fp64":
                raise ValueError("the Gauss-Hermite link is built for Cholesky q(u) covariances in fp64")
        elif linkFunction != ExponentialLink:
            raise ValueError("Invalid linkFunction={:s}".format(repr(linkFunction)))
        else:
            linkFunctionCallable = None
        if indPointsCovRep not in (indPointsCovChol, indPointsCovRank1PlusDiag):
            raise ValueError("Invalid indPointsCovRep")
        if kernelMatrixInvMethod not in (kernelMatrixInvChol, kernelMatrixInvPInv):
            raise ValueError("Invalid kernelMatrixInvMethod")
        model = svLowerBound.SVLowerBound(device=device, rank1_plus_diag=(indPointsCovRep == indPointsCovRank1PlusDiag),
                                          precision=precision, link=linkFunctionCallable)
        model.setKernels(kernels=kernels)
        return model
