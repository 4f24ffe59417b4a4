(* GPHIP.wl -- thin Wolfram-Language host package for the MI355X GP path.

   Load AFTER the reference package (BayesianInference/Kernel/BayesianInference.wl:11-19): it reuses the
   reference's own inferenceObject, $MachineLogZero, dataNormalForm and defineInferenceProblem and swaps
   (1) the "LogLikelihoodFunction" closure (seam at BayesianGaussianProcess.wl:249, 293-294) and
   (2) the prediction down-value for HIP-backed objects (BayesianGaussianProcess.wl:343-376),
   keeping every key of "GaussianProcessData"/"ModelFunctions" in the SHAPE the reference defines
   (BayesianGaussianProcess.wl:257-262, 308, 314-321) so that reference code reading the object --
   predictFromGaussianProcess' own loop (:358-368), regressionPlot1D -- keeps working on it:
       "KernelFunction", "NuggetFunction", "MeanFunction"   theta |-> pure function   (expressionToFunction, :257-262)
       "CovarianceFunction"                                 theta |-> N x N matrix    (compiledCovarianceMatrix, :265-270)
       "InverseCovarianceFunction"                          theta |-> <|"Inverse" -> solver, "LogDet" -> real|>  (:137-141, 308)
   All numerics are in libgphip (include/gphip.h) behind the LibraryLink shim (csrc/librarylink_shim.cpp).
   This file cannot be executed in the build containers (no Wolfram kernel).  What IS tested there:
   the shim is compiled against a stub WolframLibrary.h and every gphip_wl_* entry point is driven through a
   fake WolframLibraryData on the GPU (tests/test_gpu_wl_shim.py); tests/test_wl_package.py checks that every
   LibraryFunctionLoad below names an exported shim function with the same argument count; and every call made
   here is mirrored 1:1 by bayesianinference_amd/gaussian_process.py, which the parity tests exercise. *)

BeginPackage["GPHIP`", {"BayesianUtilities`", "BayesianStatistics`", "BayesianGaussianProcess`"}]

defineGaussianProcessHIP::usage = "defineGaussianProcessHIP[X -> Y, kernelName, variables, prior, opts] builds the same inferenceObject as defineGaussianProcess with the log-likelihood evaluated on the GPU. kernelName is \"SE\", \"SEARD\", \"Matern52\", \"Matern52ARD\" or None (null kernel); variables = {{l.., min, max}.., {sf,..}, {sn,..}[, {mu,..}]}. Options: \"ConstantMean\" -> False, \"Precision\" -> \"Double\" | \"Single\", \"Devices\" -> Automatic | {0, 1, ..}, \"LibraryOptions\" -> {\"panel\" -> 4, ..}.";
hipKernelFunction::usage = "hipKernelFunction[kernelName, d] gives theta |-> Function[{p, q}, ..], the exact WL form of the named kernel (what one would hand to the reference's defineGaussianProcess for the same model).";
$GPHIPLibrary::usage = "Path of the LibraryLink shim (libgphip_wl).";

Begin["`Private`"]

$GPHIPLibrary = FindLibrary["libgphip_wl"];
kernelIds = <|"SE" -> 0, "SEARD" -> 1, "Matern52" -> 2, "Matern52ARD" -> 3, None -> 4|>;
nLengthScales[name_, d_] := Switch[name, "SE" | "Matern52", 1, "SEARD" | "Matern52ARD", d, _, 0];

(* ---- LibraryLink bindings (argument lists are checked against the shim by tests/test_wl_package.py) ---- *)
gpCreate   := gpCreate   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_create",
	{{Real, 2, "Constant"}, {Real, 1, "Constant"}, Integer, Integer, Integer, {Integer, 1, "Constant"}}, Integer];
gpSetOpt   := gpSetOpt   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_set_option", {Integer, "UTF8String", Real}, Integer];
gpLogLik   := gpLogLik   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik",
	{Integer, {Real, 1, "Constant"}}, {Real, 1}];        (* {value, info} *)
gpLogLikB  := gpLogLikB  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik_batch",
	{Integer, {Real, 2, "Constant"}}, {Real, 2}];        (* {{value, info}..} *)
gpGrad     := gpGrad     = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik_grad",
	{Integer, {Real, 1, "Constant"}}, {Real, 1}];        (* {value, info, grad..} *)
gpFit      := gpFit      = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_fit",
	{Integer, {Real, 1, "Constant"}}, Integer];          (* info *)
gpSolve    := gpSolve    = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_solve",
	{Integer, {Real, _, "Constant"}}, {Real, _}];        (* vector (N) or matrix (N x m), same shape back *)
gpLogDet   := gpLogDet   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_logdet", {Integer}, Real];
gpPredict  := gpPredict  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict",
	{Integer, {Real, 2, "Constant"}}, {Real, 2}];        (* {means, variances} *)
gpPredictS := gpPredictS = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict_samples",
	{Integer, {Real, 2, "Constant"}, {Real, 2, "Constant"}}, {Real, 3}];   (* {means, variances}, each S x M *)
gpCov      := gpCov      = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_covariance",
	{Integer, {Real, 1, "Constant"}}, {Real, 2}];
gpCross    := gpCross    = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_cross_covariance",
	{Integer, {Real, 1, "Constant"}, {Real, 2, "Constant"}}, {Real, 2}];   (* (N+1) x M: k on top, kappa last row *)
gpDestroy  := gpDestroy  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_destroy", {Integer}, Integer];

(* value -> machine real; info != 0 or a LibraryFunctionError -> $MachineLogZero, exactly what
   Catch[..., "MatInv"] yields in the reference closure (BayesianGaussianProcess.wl:298-304). *)
toLogLik[{val_Real, info_Real}] := If[info == 0., Clip[val, {-Abs[$MachineLogZero], Abs[$MachineLogZero]}], $MachineLogZero];
toLogLik[_] := $MachineLogZero;

(* ---- the named kernels as the reference would see them: theta |-> pure function ---- *)
hipKernelFunction["SE", d_] := Function[theta,
	With[{l = theta[[1]], sf = theta[[2]]},
		Function[{p, q}, sf^2 Exp[-Total[(p - q)^2]/(2 l^2)]]]];
hipKernelFunction["SEARD", d_] := Function[theta,
	With[{ls = theta[[;; d]], sf = theta[[d + 1]]},
		Function[{p, q}, sf^2 Exp[-Total[((p - q)/ls)^2]/2]]]];
hipKernelFunction["Matern52", d_] := Function[theta,
	With[{l = theta[[1]], sf = theta[[2]]},
		Function[{p, q}, With[{s = Sqrt[Total[(p - q)^2]]/l}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]]];
hipKernelFunction["Matern52ARD", d_] := Function[theta,
	With[{ls = theta[[;; d]], sf = theta[[d + 1]]},
		Function[{p, q}, With[{s = Sqrt[Total[((p - q)/ls)^2]]}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]]];
hipKernelFunction[None, d_] := Function[theta, Function[0]];      (* nullKernelPattern, BayesianGaussianProcess.wl:25 *)
hipNuggetFunction[name_, d_] := With[{i = If[name === None, 1, nLengthScales[name, d] + 2]},
	Function[theta, With[{sn = theta[[i]]}, Function[sn^2]]]];
hipMeanFunction[name_, d_, False] := Function[theta, Function[0]];
hipMeanFunction[name_, d_, True] := With[{i = If[name === None, 2, nLengthScales[name, d] + 3]},
	Function[theta, With[{mu = theta[[i]]}, Function[mu]]]];

(* the handle keeps ONE factor resident: refit only when theta changed since the last fit.  Every other call into the
   library reuses the handle's workspace, so it first forgets the fit (touch). *)
$fitted = <||>;
touch[h_] := ($fitted[h] = None);
ensureFit[h_, theta_] := If[ Lookup[$fitted, h, None] === theta,
	0,
	With[{info = gpFit[h, theta]},
		$fitted[h] = If[info === 0, theta, None];
		info
	]
];

Options[defineGaussianProcessHIP] = {"ConstantMean" -> False, "Precision" -> "Double", "Devices" -> Automatic, "LibraryOptions" -> {}};

defineGaussianProcessHIP[
	dataIn_List?(MatrixQ[#, NumericQ]&) -> dataOut_List?(MatrixQ[#, NumericQ]&),
	kernelName : (_String | None),
	variables : {{_Symbol, _, _}..},
	variablePrior_,
	rest___Rule
] /; Dimensions[dataOut][[2]] === 1 && Length[dataIn] === Length[dataOut] && KeyExistsQ[kernelIds, kernelName] := Module[{
	h, loglik, invCov,
	d = Dimensions[dataIn][[2]],
	constMean = TrueQ[Lookup[{rest}, "ConstantMean", False]],
	dtype = If[Lookup[{rest}, "Precision", "Double"] === "Single", 32, 64],
	(* sub-kernels of parallelNestedSampling pick their own GPU (BayesianStatistics.wl:1349); a list of several
	   ordinals makes ONE multi-device handle: the library shards a large factorisation over them *)
	devices = Replace[Lookup[{rest}, "Devices", Automatic], {Automatic :> {Mod[$KernelID, 8]}, i_Integer :> {i}}],
	own = "ConstantMean" | "Precision" | "Devices" | "LibraryOptions"
},
	h = gpCreate[N @ dataIn, N @ Flatten[dataOut], kernelIds[kernelName], Boole[constMean], dtype, devices];
	If[ !IntegerQ[h] || h < 0, Return[inferenceObject[$Failed]]];
	KeyValueMap[gpSetOpt[h, #1, N[#2]]&, Association @ Lookup[{rest}, "LibraryOptions", {}]];
	loglik = Function[theta,
		touch[h];
		If[ MatrixQ[theta], toLogLik /@ gpLogLikB[h, N @ theta], toLogLik @ gpLogLik[h, N @ theta]]
	];
	(* matrixInverseAndDet[covarianceFunction[theta]] (BayesianGaussianProcess.wl:130-141, 308): an Association with
	   a solver that takes a vector or a matrix (:194, :410, :416) and the log-determinant; singular K Throws the
	   sentinel with tag "MatInv" exactly like :133 *)
	invCov = Function[theta,
		With[{th = N @ theta},
			If[ ensureFit[h, th] =!= 0, Throw[$MachineLogZero, "MatInv"]];
			<|
				"Inverse" -> Function[b, If[ensureFit[h, th] =!= 0, Throw[$MachineLogZero, "MatInv"]]; gpSolve[h, N @ b]],
				"LogDet" -> gpLogDet[h]
			|>
		]
	];
	defineInferenceProblem[                                (* same keys as BayesianGaussianProcess.wl:310-325 *)
		"Data" -> dataNormalForm[dataIn -> dataOut],
		"PriorDistribution" -> variablePrior,
		"Parameters" -> variables,
		"GaussianProcessData" -> <|
			"ModelFunctions" -> <|
				"KernelFunction" -> hipKernelFunction[kernelName, d],
				"NuggetFunction" -> hipNuggetFunction[kernelName, d],
				"MeanFunction" -> hipMeanFunction[kernelName, d, constMean],
				"CovarianceFunction" -> Function[theta, touch[h]; gpCov[h, N @ theta]],
				"InverseCovarianceFunction" -> invCov
			|>,
			"KernelName" -> kernelName,
			"HIPHandle" -> h
		|>,
		Sequence @@ FilterRules[{rest}, Except[own]],
		"LogLikelihoodGradientFunction" -> Function[theta, touch[h]; gpGrad[h, N @ theta]],
		"LogLikelihoodFunction" -> loglik
	]
];
defineGaussianProcessHIP[___] := inferenceObject[$Failed];

(* ---- prediction for HIP-backed objects: same return shape as BayesianGaussianProcess.wl:343-376 ---- *)
hipObjectQ = Function[AssociationQ[#] && KeyExistsQ[#, "Samples"] &&
	KeyExistsQ[Lookup[#, "GaussianProcessData", <||>], "HIPHandle"]];

hipPredict[result_, pts_List] := Module[{
	h = result["GaussianProcessData", "HIPHandle"],
	points = dataNormalForm[pts],
	weights = Values @ result[["Samples", All, "CrudePosteriorWeight"]],
	perSample
},
	(* one batched call: every posterior sample is factored and solved in its own workspace slot; a sample whose K
	   is singular comes back as NaN rows *)
	perSample = With[{mv = gpPredictS[h, N @ Values @ result[["Samples", All, "Point"]], N @ points]},
		MapThread[
			Function[{mus, vars}, MapThread[NormalDistribution, {mus, Sqrt[vars]}]],
			{mv[[1]], mv[[2]]}
		]
	];
	touch[h];                                                (* the batched pass reused the handle's workspace *)
	AssociationThread[points, MixtureDistribution[weights, #]& /@ Transpose[perSample]]
];

(* The reference's own definition (BayesianGaussianProcess.wl:343-346) matches a HIP object just as well -- its LHS
   differs from ours only inside a PatternTest, which WL cannot order by specificity -- and it was defined first.
   So the HIP rule is PREPENDED to the down-values instead of appended by an ordinary definition.  (With the
   reference-shaped "ModelFunctions" above, the reference's rule would still work on a HIP object: S interpreted
   kernel builds + LU factorisations instead of one batched GPU pass.) *)
Unprotect[predictFromGaussianProcess];
DownValues[predictFromGaussianProcess] = Prepend[
	DownValues[predictFromGaussianProcess],
	HoldPattern[predictFromGaussianProcess[inferenceObject[result_?hipObjectQ], pts_List]] :> hipPredict[result, pts]
];

(* predictiveDistribution (BayesianStatistics.wl:1373-1387) needs a "GeneratingDistribution", which a GP object does
   not carry; for HIP-backed GP objects it forwards to the prediction above.  The "MaximumLikelihood" / "MAP"
   forms (:1389-1416) reduce "Samples" to one element and re-enter here. *)
bestSample[result_, f_] := Append[result, "Samples" -> TakeLargestBy[result["Samples"], f, 1]];
Unprotect[predictiveDistribution];
DownValues[predictiveDistribution] = Join[
	{
		HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List]] :>
			hipPredict[result, pts],
		HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List, "MaximumLikelihood"]] :>
			hipPredict[bestSample[result, #LogLikelihood &], pts],
		HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List, "MAP"]] :>
			hipPredict[bestSample[result, #LogLikelihood + #LogPriorPDF &], pts]
	},
	DownValues[predictiveDistribution]
];

End[]
EndPackage[]
