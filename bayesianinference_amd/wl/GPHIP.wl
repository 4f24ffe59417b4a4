(* GPHIP.wl -- thin Wolfram-Language host package for the MI355X GP path.

   Load AFTER the reference package (BayesianInference/Kernel/BayesianInference.wl:11-19): it reuses
   the reference's own inferenceObject, $MachineLogZero, dataNormalForm and defineInferenceProblem
   and only swaps the "LogLikelihoodFunction" closure (seam at BayesianGaussianProcess.wl:249,
   293-294) and the prediction down-value (BayesianGaussianProcess.wl:343-376).
   All numerics are in libgphip (include/gphip.h) behind the LibraryLink shim
   (csrc/librarylink_shim.cpp).  This file cannot be executed in the build containers (no Wolfram
   kernel); every call it makes is mirrored 1:1 by bayesianinference_amd/gaussian_process.py, which
   is what the tests exercise. *)

BeginPackage["GPHIP`", {"BayesianUtilities`", "BayesianStatistics`", "BayesianGaussianProcess`"}]

defineGaussianProcessHIP::usage = "defineGaussianProcessHIP[X -> Y, kernelName, variables, prior, opts] builds the same inferenceObject as defineGaussianProcess with the log-likelihood evaluated on the GPU. kernelName is \"SE\", \"SEARD\", \"Matern52\" or \"Matern52ARD\"; variables = {{l.., min, max}.., {sf,..}, {sn,..}[, {mu,..}]}.";
$GPHIPLibrary::usage = "Path of the LibraryLink shim (libgphip_wl).";

Begin["`Private`"]

$GPHIPLibrary = FindLibrary["libgphip_wl"];
kernelIds = <|"SE" -> 0, "SEARD" -> 1, "Matern52" -> 2, "Matern52ARD" -> 3, None -> 4|>;

gpCreate  := gpCreate  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_create",
	{{Real, 2, "Constant"}, {Real, 1, "Constant"}, Integer, Integer, Integer}, Integer];
gpLogLik  := gpLogLik  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik",
	{Integer, {Real, 1, "Constant"}}, {Real, 1}];        (* {value, info} *)
gpLogLikB := gpLogLikB = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik_batch",
	{Integer, {Real, 2, "Constant"}}, {Real, 2}];        (* {{value, info}..} *)
gpFit     := gpFit     = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_fit",
	{Integer, {Real, 1, "Constant"}}, Integer];
gpPredict := gpPredict = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict",
	{Integer, {Real, 2, "Constant"}}, {Real, 2}];        (* {means, variances} *)
gpPredictS := gpPredictS = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict_samples",
	{Integer, {Real, 2, "Constant"}, {Real, 2, "Constant"}}, {Real, 3}];   (* {means, variances}, each S x M *)
gpCov     := gpCov     = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_covariance",
	{Integer, {Real, 1, "Constant"}}, {Real, 2}];
gpDestroy := gpDestroy = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_destroy", {Integer}, Integer];

(* value -> machine real; info != 0 or a LibraryFunctionError -> $MachineLogZero, exactly what
   Catch[..., "MatInv"] yields in the reference closure (BayesianGaussianProcess.wl:298-304). *)
toLogLik[{val_Real, info_Real}] := If[info == 0., Clip[val, {-Abs[$MachineLogZero], Abs[$MachineLogZero]}], $MachineLogZero];
toLogLik[_] := $MachineLogZero;

defineGaussianProcessHIP[
	dataIn_List?(MatrixQ[#, NumericQ]&) -> dataOut_List?(MatrixQ[#, NumericQ]&),
	kernelName : (_String | None),
	variables : {{_Symbol, _, _}..},
	variablePrior_,
	rest___Rule
] /; Dimensions[dataOut][[2]] === 1 && Length[dataIn] === Length[dataOut] := Module[{
	h, meanId = If[TrueQ[Lookup[{rest}, "ConstantMean", False]], 1, 0], loglik
},
	h = gpCreate[N @ dataIn, N @ Flatten[dataOut], kernelIds[kernelName], meanId,
		Lookup[{rest}, "Device", Mod[$KernelID, 8]]];      (* sub-kernels pick their own GPU, BS:1349 *)
	If[ !IntegerQ[h] || h < 0, Return[inferenceObject[$Failed]]];
	loglik = Function[theta,
		If[ MatrixQ[theta], toLogLik /@ gpLogLikB[h, N @ theta], toLogLik @ gpLogLik[h, N @ theta]]
	];
	defineInferenceProblem[                                (* same keys as BayesianGaussianProcess.wl:310-325 *)
		"Data" -> dataNormalForm[dataIn -> dataOut],
		"PriorDistribution" -> variablePrior,
		"Parameters" -> variables,
		"GaussianProcessData" -> <|
			"ModelFunctions" -> <|
				"KernelFunction" -> kernelName,
				"NuggetFunction" -> "Constant",
				"MeanFunction" -> meanId,
				"CovarianceFunction" -> Function[gpCov[h, N @ #]],
				"InverseCovarianceFunction" -> Function[gpFit[h, N @ #]]
			|>,
			"HIPHandle" -> h
		|>,
		Sequence @@ FilterRules[{rest}, Except["Device" | "ConstantMean"]],
		"LogLikelihoodFunction" -> loglik
	]
];
defineGaussianProcessHIP[___] := inferenceObject[$Failed];

(* prediction for HIP-backed objects: same return shape as BayesianGaussianProcess.wl:343-376 *)
Unprotect[predictFromGaussianProcess];
predictFromGaussianProcess[
	inferenceObject[result_?(AssociationQ[#] && KeyExistsQ[#, "Samples"] &&
		KeyExistsQ[Lookup[#, "GaussianProcessData", <||>], "HIPHandle"]&)],
	pts_List
] := Module[{
	h = result["GaussianProcessData", "HIPHandle"],
	points = dataNormalForm[pts],
	weights = Values @ result[["Samples", All, "CrudePosteriorWeight"]],
	perSample
},
	(* one batched call: every posterior sample is factored and solved in its own workspace slot *)
	perSample = With[{mv = gpPredictS[h, N @ Values @ result[["Samples", All, "Point"]], N @ points]},
		MapThread[
			Function[{mus, vars}, MapThread[NormalDistribution, {mus, Sqrt[vars]}]],
			{mv[[1]], mv[[2]]}
		]
	];
	AssociationThread[points, MixtureDistribution[weights, #]& /@ Transpose[perSample]]
];

End[]
EndPackage[]
